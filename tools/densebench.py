#!/usr/bin/env python3
"""Config-5 timing: 60 tags / 240 corners + 61,440 dense samples, 1280x720 (development aid, GPU box only)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from accurate_aprilgroup_tracking_amd import cv_hip, synthetic as syn
s = syn.Sequence(1280, 720, n_tags=60, n_frames=2, seed=8, supersample=2)
mx = syn.model_samples(s.group, 32)
T = np.nan_to_num(syn.sample_bilinear(s.frame(1), syn.project(mx, s.rvecs[1], s.tvecs[1], s.K)), nan=128.0).astype(np.float32)
ctx = cv_hip.Context(64, 64, max_level=0)
for B in (1, 16):
    frames = torch.from_numpy(np.stack([s.frame(1)] * B)).cuda()
    mxd, Td = torch.from_numpy(mx).cuda(), torch.from_numpy(T).cuda()
    obj = torch.from_numpy(s.obj.astype(np.float32)).cuda(); ip = torch.from_numpy(np.stack([s.corners(1)] * B)).cuda().contiguous()
    start = torch.from_numpy(np.repeat(np.concatenate([s.rvecs[1] + 0.002, s.tvecs[1] - 0.0006])[None], B, 0)).cuda()
    ITERS = 5
    def run():
        pose = start.clone()
        ctx.dense_refine(frames, mxd, Td, pose, s.K, None, obj=obj, img_pts=ip, iters=ITERS, photo_weight=0.05)
        return pose
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    print("B=%-2d dense refine, %d GN iterations: %.1f us/call = %.1f us per iteration (2 launches), %.0f refinements/s" % (B, ITERS, us, us / ITERS, B * 1e6 / us))
