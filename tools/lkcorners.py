#!/usr/bin/env python3
"""Per-corner residence of ONE one-wave-per-corner LK launch on a 64 x 1280x720 batch (BASELINE configs[2] geometry), from the
diagnostic library's per-corner log (entry / exit s_memtime of every corner): when do waves start, how long do they live, who is last.
LKB streams (default 64); AGT_LK_WIDE_MAX=0 is not needed (3,072 corners take the one-wave kernel)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from accurate_aprilgroup_tracking_amd import hiplib
hiplib.LIB_PATH = os.path.join(os.path.dirname(hiplib.LIB_PATH), "libagt_hip_dbg.so")
from accurate_aprilgroup_tracking_amd import synthetic as syn, cv_hip
W, H, B = 1280, 720, int(os.environ.get("LKB", "64"))
seqs = [syn.Sequence(W, H, n_frames=3, seed=s, supersample=2, group_seed=0) for s in range(4)]
ctx = cv_hip.Context(W, H, max_level=2, max_points=48, max_streams=B)
f = [torch.from_numpy(np.stack([seqs[b % 4].frame(k) for b in range(B)])).cuda().contiguous() for k in range(3)]
ctx.pyramid_build(0, f[0]); ctx.pyramid_build(1, f[1])
pts = torch.from_numpy(np.stack([seqs[b % 4].corners(0) for b in range(B)])).cuda().contiguous()
nx, st, er = ctx.lk_track(0, 1, pts)
L = hiplib.lib()
n = B * 48
for rep in range(3):
    ctx.lk_track(0, 1, pts, nx, want_err=False); torch.cuda.synchronize()
    buf = (C.c_ulonglong * (3 * n))(); L.agt_debug_lk_corner_log(buf, n)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 3).astype(np.int64)
    t0 = a[:, 0].min()
    start = (a[:, 0] - t0) / 2100.0; end = (a[:, 1] - t0) / 2100.0      # s_memtime counts shader cycles (~2.1 GHz under load; every XCD has its own base: only differences within a wave mean anything)
    life = end - start
    q = lambda v, p: float(np.percentile(v, p))
    print("rep %d: %d corners | start us: median %.2f p90 %.2f max %.2f | life us: min %.2f median %.2f p90 %.2f p99 %.2f max %.2f | last exit %.2f us"
          % (rep, n, q(start, 50), q(start, 90), start.max(), life.min(), q(life, 50), q(life, 90), q(life, 99), life.max(), end.max()))
    it = a[:, 2]
    late = np.argsort(-life)[:10]
    print("   longest-lived corners (stream:corner life us, iterations):", ", ".join("%d:%d %.1f, %d" % (c // 48, c % 48, life[c], it[c]) for c in late))
    print("   iterations: median %d p90 %d max %d | life / iteration fit: %.2f us + %.3f us per iteration" % ((np.median(it), np.percentile(it, 90), it.max()) + tuple(np.polyfit(it, life, 1)[::-1])))
    for lo, hi in ((0, 8), (8, 12), (12, 16), (16, 24), (24, 200)):
        m = (it >= lo) & (it < hi)
        if m.any(): print("      %3d..%3d iterations: %4d corners, life median %.1f max %.1f" % (lo, hi - 1, m.sum(), np.median(life[m]), life[m].max()))
    h, edges = np.histogram(life, bins=np.arange(0, life.max() + 2, 2.0))
    print("   life histogram (2-us bins from 0):", " ".join(str(int(v)) for v in h))
