#!/usr/bin/env python3
"""development aid: the changing-frame-pitch stream of tests/test_gpu_tracker.py at several depths, first differing record"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from accurate_aprilgroup_tracking_amd import synthetic as syn
from accurate_aprilgroup_tracking_amd.tracker import StreamTracker
s = syn.Sequence(640, 480, n_tags=12, n_frames=6, seed=0)
frames = torch.from_numpy(s.frames()).cuda()
H, W = s.height, s.width
order = [1, 2, 3, 4, 5, 4, 3, 2, 1, 0, 1, 2, 3, 4, 5, 4, 3, 2, 1, 0, 1, 2, 3]
padded = torch.zeros((len(order), 1, H, W + 64), dtype=torch.uint8, device="cuda")
outs = []
for depth in [int(a) for a in sys.argv[1:]] or [0, 1, 2, 4]:
    trk = StreamTracker(W, H, s.obj, s.K, None, n_streams=1)
    trk.pipeline(depth)
    trk.reset(frames[0:1].contiguous(), torch.from_numpy(s.corners(0)[None]).cuda().contiguous())
    so = trk.new_state_buffer(len(order))
    for i, k in enumerate(order):
        if i % 3 == 1 or i in (10, 11, 12, 13, 14):
            padded[i, 0, :, :W] = frames[k]
            f = padded[i, :, :, :W]
        else:
            f = frames[k:k + 1]
        trk.step(f, so[i])
    trk.join()
    outs.append(so.cpu().numpy())
    d = np.argwhere(outs[0] != outs[-1])
    print("depth", depth, "equal" if len(d) == 0 else "first differing (frame, stream, field): %s; frames %s" % (d[:4].tolist(), sorted(set(d[:, 0].tolist()))))
    if len(d):
        f0 = d[0, 0]; print("  serial", outs[0][f0, 0, :13]); print("  depth ", outs[-1][f0, 0, :13])
