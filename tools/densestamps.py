#!/usr/bin/env python3
"""Timeline of one dense_accum_kernel launch (block 1 of stream 0, Gauss-Newton iteration 2) from in-kernel s_memtime stamps
(diagnostic library: make -C csrc dbg).  Config-5 sizes: 240 corners, 61,440 samples, 1280x720."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from accurate_aprilgroup_tracking_amd import hiplib
hiplib.LIB_PATH = os.path.join(os.path.dirname(hiplib.LIB_PATH), "libagt_hip_dbg.so")
from accurate_aprilgroup_tracking_amd import cv_hip, synthetic as syn
s = syn.Sequence(1280, 720, n_tags=60, n_frames=2, seed=8, supersample=2)
mx = syn.model_samples(s.group, 32)
T = np.nan_to_num(syn.sample_bilinear(s.frame(1), syn.project(mx, s.rvecs[1], s.tvecs[1], s.K)), nan=128.0).astype(np.float32)
ctx = cv_hip.Context(64, 64, max_level=0)
frames = torch.from_numpy(s.frame(1)[None]).cuda()
mxd, Td = torch.from_numpy(mx).cuda(), torch.from_numpy(T).cuda()
obj = torch.from_numpy(s.obj.astype(np.float32)).cuda(); ip = torch.from_numpy(s.corners(1)[None]).cuda().contiguous()
start = torch.from_numpy(np.concatenate([s.rvecs[1] + 0.002, s.tvecs[1] - 0.0006])[None]).cuda()
L = hiplib.lib()
L.agt_debug_dense_stamps.argtypes = [C.c_void_p]
TICK = float(os.environ.get("AGT_TICKS_PER_US", "100.0"))
for rep in range(6):
    pose = start.clone()
    ctx.dense_refine(frames, mxd, Td, pose, s.K, None, obj=obj, img_pts=ip, iters=5, photo_weight=0.05)
    torch.cuda.synchronize()
    st = (C.c_ulonglong * 16)(); L.agt_debug_dense_stamps(st)
    if rep < 3: continue
    f = lambda i: (st[i] - st[0]) / TICK
    print("entry 0 | rows summed %.2f | barrier %.2f | totals read %.2f | solved %.2f | update done %.2f | rodrigues %.2f | projected %.2f | taps %.2f | products %.2f | butterfly %.2f | barrier %.2f us"
          % (f(8), f(9), f(10), f(11), f(1), f(2), f(3), f(4), f(5), f(6), f(7)))
