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
    buf = (C.c_ulonglong * (4 * n))(); L.agt_debug_lk_corner_log(buf, n)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 4).astype(np.int64)
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
    # where the waves ran (HW_ID: simd [5:4], cu [11:8], sh [12], se [15:13]; XCC_ID [3:0]): corners per SIMD, and a corner's life against the number of
    # corners its SIMD got in this launch
    hw = a[:, 3]
    simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; xcc = (hw >> 32) & 15
    key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
    uniq, inv, cnt = np.unique(key, return_inverse=True, return_counts=True)
    print("   %d SIMDs used (%d CUs, %d XCCs); corners per SIMD:" % (len(uniq), len(np.unique(key >> 2)), len(np.unique(xcc))),
          " ".join("%d:%d" % (k, int((cnt == k).sum())) for k in sorted(set(cnt.tolist()))))
    for k in sorted(set(cnt.tolist())):
        m = cnt[inv] == k
        print("      corners on a SIMD that got %d: %4d corners, life median %.1f p90 %.1f max %.1f" % (k, int(m.sum()), np.median(life[m]), np.percentile(life[m], 90), life[m].max()))
    percu = np.unique(key >> 2, return_counts=True)[1]
    print("   corners per CU: min %d median %d max %d" % (percu.min(), int(np.median(percu)), percu.max()))
    for x in sorted(set(xcc.tolist())):
        m = xcc == x
        print("      XCC %d: %4d corners, life median %.1f max %.1f" % (x, int(m.sum()), np.median(life[m]), life[m].max()))
    # the three corners of a SIMD, ranked by life: does the SIMD's issue arbiter serve them in order?
    if (cnt == 3).all():
        order = np.argsort(key, kind="stable")
        trip = np.sort(life[order].reshape(-1, 3), axis=1)
        wave = ((hw >> 0) & 15)[order].reshape(-1, 3)
        print("   per SIMD, lives ranked: shortest median %.1f, middle %.1f, longest %.1f us (sum %.1f); start spread within a SIMD median %.2f us"
              % (np.median(trip[:, 0]), np.median(trip[:, 1]), np.median(trip[:, 2]), np.median(trip.sum(axis=1)),
                 np.median(np.ptp(start[order].reshape(-1, 3), axis=1))))

