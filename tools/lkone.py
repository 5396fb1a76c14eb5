#!/usr/bin/env python3
"""One 1280x720 stream, 48 corners: the stand-alone LK launch with four waves per corner (default) against one wave per
corner (AGT_LK_WIDE_MAX=0, diagnostic library only).  K launches back to back, HIP events."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from accurate_aprilgroup_tracking_amd import hiplib
hiplib.LIB_PATH = os.path.join(os.path.dirname(hiplib.LIB_PATH), "libagt_hip_dbg.so")
from accurate_aprilgroup_tracking_amd import synthetic as syn, cv_hip
W, H = 1280, 720
seq = syn.Sequence(W, H, n_frames=2, seed=0, supersample=2)
ctx = cv_hip.Context(W, H, max_level=2, max_points=48, max_streams=1)
f0 = torch.from_numpy(seq.frame(0)).cuda().unsqueeze(0).contiguous(); f1 = torch.from_numpy(seq.frame(1)).cuda().unsqueeze(0).contiguous()
ctx.pyramid_build(0, f0); ctx.pyramid_build(1, f1)
pts = torch.from_numpy(seq.corners(0)[None]).cuda().contiguous()
nx, st, er = ctx.lk_track(0, 1, pts)
K = 200
for _ in range(20): ctx.lk_track(0, 1, pts, nx, want_err=False)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(K): ctx.lk_track(0, 1, pts, nx, want_err=False)
e1.record(); torch.cuda.synchronize()
print("AGT_LK_WIDE_MAX=%s  lk %.2f us/launch" % (os.environ.get("AGT_LK_WIDE_MAX", "default"), e0.elapsed_time(e1) / K * 1e3))
