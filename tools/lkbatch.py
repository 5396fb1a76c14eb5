#!/usr/bin/env python3
"""The one-wave-per-corner LK kernel alone on a 64 x 1280x720 batch (BASELINE configs[2] geometry): K launches back to back.
Used under rocprofv3 --pmc (SQ counters per wave) and --kernel-trace; distinct frames per stream so the tiles come from HBM / MALL."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from accurate_aprilgroup_tracking_amd import synthetic as syn, cv_hip

W, H, B = 1280, 720, int(os.environ.get("LKB", "64"))
K = int(os.environ.get("LKK", "20"))
seqs = [syn.Sequence(W, H, n_frames=2, seed=s, supersample=2, group_seed=0) for s in range(4)]
ctx = cv_hip.Context(W, H, max_level=2, max_points=48, max_streams=B)
f0 = torch.from_numpy(np.stack([seqs[b % 4].frame(0) for b in range(B)])).cuda().contiguous()
f1 = torch.from_numpy(np.stack([seqs[b % 4].frame(1) for b in range(B)])).cuda().contiguous()
ctx.pyramid_build(0, f0); ctx.pyramid_build(1, f1)
pts = torch.from_numpy(np.stack([seqs[b % 4].corners(0) for b in range(B)])).cuda().contiguous()
nx, st, er = ctx.lk_track(0, 1, pts)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(K):
    ctx.lk_track(0, 1, pts, nx, want_err=False)
e1.record(); torch.cuda.synchronize()
print("B=%d  lk %.2f us/launch  tracked %d / %d" % (B, e0.elapsed_time(e1) / K * 1e3, int(st.sum()), st.numel()))
