#!/usr/bin/env python3
"""Timeline of one lk_kernel wave from in-kernel s_memtime stamps (diagnostic build: make -C csrc dbg)."""
import ctypes as C, os, sys
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
L = hiplib.lib()
for rep in range(3):
    ctx.lk_track(0, 1, pts); torch.cuda.synchronize()
    st = (C.c_ulonglong * 64)(); L.agt_debug_lk_stamps(st)
    t0 = st[0]
    f = lambda i: (st[i] - t0) / 2100.0          # s_memtime counts shader cycles (~2.1 GHz under this load): us
    print("rep %d: loads issued %.2f us, tiles in LDS %.2f us, end %.2f us" % (rep, f(1), f(2), f(3)))
    for lv in (2, 1, 0):
        b = 8 + lv * 8
        print("   level %d: start %.2f  scharr+%.2f  patch+sums+%.2f  iter0+%.2f  iters(n=%d)+%.2f" % (
            lv, f(b), f(b + 1) - f(b), f(b + 2) - f(b + 1), f(b + 3) - f(b + 2), st[b + 6], f(b + 4) - f(b + 2)))
    if st[45] > st[39] > 0:       # per-iteration stamps exist only in the general body (AGT_LK_RS=0)
        g = lambda i: st[i] - st[39]
        print("   iteration j=1 of the last level run (cycles): floor/bounds %d | weights %d | window pass %d | block sum %d | delta %d | conds %d" % (
            g(40), g(41) - g(40), g(42) - g(41), g(43) - g(42), g(44) - g(43), g(45) - g(44)))
