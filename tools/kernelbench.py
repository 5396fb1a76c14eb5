#!/usr/bin/env python3
"""Per-kernel back-to-back timing (development aid, GPU box only): each kernel launched K times
in a row on the same stream, so launches see warm caches; compare with the in-pipeline spans."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from accurate_aprilgroup_tracking_amd import synthetic as syn, cv_hip

W, H = 1280, 720
seq = syn.Sequence(W, H, n_frames=2, seed=0, supersample=2)
for B in (16, 32, 64, 128):
    ctx = cv_hip.Context(W, H, max_level=2, max_points=48, max_streams=B)
    f0 = torch.from_numpy(seq.frame(0)).cuda().unsqueeze(0).expand(B, H, W).contiguous()
    f1 = torch.from_numpy(seq.frame(1)).cuda().unsqueeze(0).expand(B, H, W).contiguous()
    ctx.pyramid_build(0, f0); ctx.pyramid_build(1, f1)
    pts = torch.from_numpy(np.repeat(seq.corners(0)[None], B, 0)).cuda().contiguous()
    obj = torch.from_numpy(seq.obj.astype(np.float32)).cuda()
    guess = np.concatenate([seq.rvecs[0], seq.tvecs[0]])
    K = 200

    def timeit(fn, name, nbytes=None):
        for _ in range(10): fn()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(K): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / K * 1e3
        extra = "  %.1f GB/s" % (nbytes / us / 1e3) if nbytes else ""
        print("B=%-3d %-14s %8.2f us/launch%s" % (B, name, us, extra))

    timeit(lambda: ctx.pyr_down(f0), "pyr_down L0", B * (W * H * 1.25))
    nx, st, er = ctx.lk_track(0, 1, pts)
    timeit(lambda: ctx.lk_track(0, 1, pts, nx, want_err=False), "lk")
    img = nx.clone()
    pose = torch.from_numpy(np.repeat(guess[None], B, 0)).cuda().contiguous()

    def pnp():
        pose.copy_(pose0)
        ctx.solve_pnp(obj, img, seq.K, None, pose, True)
    pose0 = pose.clone()
    timeit(pnp, "pnp(+copy)")
    timeit(lambda: pose.copy_(pose0), "copy only")
