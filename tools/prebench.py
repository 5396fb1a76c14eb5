#!/usr/bin/env python3
"""fused undistort + gray + crop kernel timing on 1280x720 BGR frames (development aid, GPU box only)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from accurate_aprilgroup_tracking_amd import cv_hip, synthetic as syn
W, H = 1280, 720
K = syn.camera_matrix(W, H); dist = np.array([[-0.25, 0.1, 1e-3, -5e-4, -0.02]])
newK, roi = cv_hip.getOptimalNewCameraMatrix(K, dist, (W, H), 1, (W, H))
for B in (1, 64):
    ctx = cv_hip.Context(64, 64, max_level=0)
    ctx.undistort_init(K, dist, newK, W, H)
    SL = max(2, (300 << 20) // (B * W * H * 3))
    src = torch.randint(0, 256, (SL, B, H, W, 3), dtype=torch.uint8, device="cuda")
    out = ctx.preprocess_bgr(src[0], roi)
    n = 50
    for mode, und in (("undistort+gray+crop", True), ("gray+crop only", False)):
        for i in range(5): ctx.preprocess_bgr(src[i % SL], roi, undistort=und, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n): ctx.preprocess_bgr(src[i % SL], roi, undistort=und, out=out)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        byt = B * (roi[2] * roi[3] * (3 + 1) + (6 * roi[2] * roi[3] if und else 0) / B)
        print("B=%-3d %-22s %8.2f us/launch  %.0f GB/s algorithmic (3 B/px in + 1 B/px out + maps once)" % (B, mode, us, byt / us / 1e3))
