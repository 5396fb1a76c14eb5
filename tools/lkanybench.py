#!/usr/bin/env python3
"""What the general LK body (any window, agt_lk_any_body.h) costs against the compiled-in ones: one agt_lk_track call on 1280x720 pairs, 48 corners
per stream, 1 and 64 streams, HIP events around 20 calls.  python tools/lkanybench.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from accurate_aprilgroup_tracking_amd import synthetic as syn, cv_hip
W, H = 1280, 720
seqs = [syn.Sequence(W, H, n_frames=2, seed=s, supersample=2, group_seed=0) for s in range(4)]
for B in (1, 64):
    f0 = torch.from_numpy(np.stack([seqs[b % 4].frame(0) for b in range(B)])).cuda().contiguous()
    f1 = torch.from_numpy(np.stack([seqs[b % 4].frame(1) for b in range(B)])).cuda().contiguous()
    pts = torch.from_numpy(np.stack([seqs[b % 4].corners(0) for b in range(B)])).cuda().contiguous()
    for name, win in (("21 compiled-in", 21), ("21 general", 21 | (21 << 8)), ("15 compiled-in", 15), ("15 general", 15 | (15 << 8)), ("31 compiled-in", 31),
                      ("31 general", 31 | (31 << 8)), ("9 general", 9), ("13 general", 13), ("17x11 general", 17 | (11 << 8)), ("45 general", 45), ("63 general", 63)):
        ctx = cv_hip.Context(W, H, max_level=2, win=win, max_points=48, max_streams=B)
        ctx.pyramid_build(0, f0); ctx.pyramid_build(1, f1)
        nx, st, er = ctx.lk_track(0, 1, pts)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ctx.lk_track(0, 1, pts, nx, want_err=False)
        e1.record(); torch.cuda.synchronize()
        print("B=%2d  %-16s %8.1f us per call, %d of %d corners tracked" % (B, name, e0.elapsed_time(e1) * 1e3 / 20, int(st.sum()), B * 48), flush=True)
        del ctx
