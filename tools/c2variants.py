#!/usr/bin/env python3
"""One 1280x720 stream through the fused chained step on the DIAGNOSTIC library (make -C csrc dbg), so that its
environment knobs apply: AGT_LK_WIDE_MAX=0 (one wave per corner in the LK role), AGT_LK_RS=0 (general LK body only),
AGT_CHAIN=0 (PnP one launch behind LK).  Prints frames/s of a 400-frame clip at the given depth (default 16)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from accurate_aprilgroup_tracking_amd import hiplib
hiplib.LIB_PATH = os.path.join(os.path.dirname(hiplib.LIB_PATH), os.environ.get("AGT_LIB", "libagt_hip_dbg.so"))
from accurate_aprilgroup_tracking_amd import synthetic as syn
from accurate_aprilgroup_tracking_amd.tracker import StreamTracker
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 16
K = int(sys.argv[2]) if len(sys.argv) > 2 else 400
NF = 24
seq = syn.Sequence(1280, 720, n_frames=NF, seed=0, supersample=2)
fr = torch.from_numpy(seq.frames()).cuda()
idx = [(i % (2 * NF - 2)) if (i % (2 * NF - 2)) < NF else 2 * NF - 2 - (i % (2 * NF - 2)) for i in range(K + 1)]
clip = fr[idx].unsqueeze(1).contiguous()            # [K+1, 1, H, W]
trk = StreamTracker(1280, 720, seq.obj, seq.K, None, n_streams=1)
if os.environ.get("AGT_REPS"):          # library built with -DAGT_CHAIN_REPS: repeat sections of the chained LK role's per-frame work
    import ctypes as C
    reps = (C.c_int * 8)(*([int(v) for v in os.environ["AGT_REPS"].split(",")] + [1] * 8)[:8])
    hiplib.lib().agt_debug_chain_reps(reps)
trk.pipeline(depth)
so = torch.zeros((K, 1, 16), dtype=torch.float64, device="cuda")
best = 1e9
for rep in range(5):
    trk.reset(clip[0], torch.from_numpy(seq.corners(0)[None]).cuda().contiguous())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    trk.step_many(clip[1:], so); trk.join(); torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t0)
st = so.cpu().numpy()
print("env %s depth %d K %d: %.1f frames/s (%.2f us/frame), accepted %.3f, iters %.2f" % (
    {k: v for k, v in os.environ.items() if k.startswith("AGT_")}, depth, K, K / best, best / K * 1e6, st[:, 0, 6].mean(), st[:, 0, 9].mean()))

try:
    import ctypes as C
    L = hiplib.lib(); cnt = (C.c_uint * 8)(); L.agt_debug_chain_counts.argtypes = [C.c_void_p]; L.agt_debug_chain_counts(cnt)
    n = max(cnt[0], 1)
    print("   chained LK role: %d corner-frames, previous-image tile reloaded in %.1f %%, search tile loaded on demand in %.1f %%, re-stages %.2f %%, iterations per corner-frame %.2f"
          % (cnt[0], 100.0 * cnt[1] / n, 100.0 * cnt[2] / n, 100.0 * cnt[3] / n, cnt[4] / n))
except Exception as e:
    print("   (no chain counters: %s)" % e)
