#!/usr/bin/env python3
"""Timeline of one pnp_kernel wave in tracker mode from in-kernel s_memtime stamps (make -C csrc dbg)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from accurate_aprilgroup_tracking_amd import hiplib
hiplib.LIB_PATH = os.path.join(os.path.dirname(hiplib.LIB_PATH), "libagt_hip_dbg.so")
from accurate_aprilgroup_tracking_amd import synthetic as syn
from accurate_aprilgroup_tracking_amd.tracker import StreamTracker
NT = int(sys.argv[1]) if len(sys.argv) > 1 else 12          # tags: 12 -> 48 corners (one wave), 60 -> 240 (four cooperating waves)
seq = syn.Sequence(1280, 720, n_tags=NT, n_frames=12, seed=0, supersample=1)
trk = StreamTracker(1280, 720, seq.obj, seq.K, None, n_streams=1)
trk.reset()
L = hiplib.lib()
so = trk.new_state_buffer()
for k in range(10):
    img = torch.from_numpy(seq.corners(k)[None]).cuda().contiguous()
    trk.estimate_pose(img, None, so); torch.cuda.synchronize()
    st = (C.c_ulonglong * 64)(); L.agt_debug_pnp_stamps(st)
    if k < 6: continue
    t0 = st[0]; f = lambda i: (st[i] - t0) / 2387.0
    it = int(so.cpu().numpy()[0, hiplib.ST_ITERS])
    print("frame %d: iters %d | loads %.2f us, first J-eval +%.2f | LM done %.2f | mean-err +%.2f | state machine +%.2f" % (
        k, it, f(1), f(2) - f(1), f(3), f(4) - f(3), f(5) - f(4)))
    for i in range(it):
        b = 8 + i * 4
        print("    iter %d: solve %.2f  err-eval %.2f  J-eval %.2f" % (i, f(b + 1) - f(b), f(b + 2) - f(b + 1), (f(b + 3) - f(b + 2)) if i < it - 1 else 0.0))
    print("    last mode-2 evaluation: rodrigues %.2f | per-point projection + products %.2f | 28-sum reduction %.2f us" % (
        (st[49] - st[48]) / 2387.0, (st[50] - st[49]) / 2387.0, (st[51] - st[50]) / 2387.0))
