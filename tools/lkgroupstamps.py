#!/usr/bin/env python3
"""The one-wave-per-corner LK role as a GROUP launch (in-kernel frame loop, knobs build: AGT_SPLIT_LK_GROUP=2) with in-kernel stamps:
timeline of corner 0's last frame, slowest corner per frame index and the mean corner-frame time over all corners.  Needs the experiment
library libagt_hip_exp.so:
    cd accurate_aprilgroup_tracking_amd/csrc && make knobs && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math \
        -DAGT_DEBUG_KNOBS -DAGT_STEP_LK_STAMPS -c agt_step.hip -o /tmp/step_st.o && hipcc --offload-arch=gfx950 -shared -fPIC -o ../libagt_hip_exp.so \
        agt_api.knobs.o agt_pyramid.knobs.o agt_lk.knobs.o agt_pnp.knobs.o /tmp/step_st.o agt_preproc.knobs.o agt_dense.knobs.o
    AGT_SPLIT_LK_GROUP=2 LKB=64 python tools/lkgroupstamps.py
Round 4: mean corner-frame 22.2 us at 24 streams, 52 us at 64 streams (a corner alone: 14.3 us with 8 iterations) -- the chip, not the launch
structure, bounds the 64-stream step."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from accurate_aprilgroup_tracking_amd import hiplib
hiplib.LIB_PATH = os.path.join(os.path.dirname(hiplib.LIB_PATH), "libagt_hip_exp.so")
from accurate_aprilgroup_tracking_amd import synthetic as syn
from accurate_aprilgroup_tracking_amd.tracker import StreamTracker
B = int(os.environ.get("LKB", "24"))
seqs = [syn.Sequence(1280, 720, n_frames=8, seed=s, supersample=2, group_seed=0) for s in range(4)]
fr = [torch.from_numpy(s.frames()).cuda() for s in seqs]
trk = StreamTracker(1280, 720, seqs[0].obj, seqs[0].K, None, n_streams=B)
trk.pipeline(16)
f0 = torch.stack([fr[b % 4][0] for b in range(B)]).contiguous()
c0 = torch.from_numpy(np.stack([seqs[b % 4].corners(0) for b in range(B)])).cuda().contiguous()
trk.reset(f0, c0)
pp = lambda i, nf=8: (i % (2 * nf - 2)) if (i % (2 * nf - 2)) < nf else 2 * nf - 2 - (i % (2 * nf - 2))
keep = []
for i in range(1, 65):
    f = torch.stack([fr[b % 4][pp(i)] for b in range(B)]).contiguous(); keep.append(f)
    trk.step(f, None)
trk.join(); torch.cuda.synchronize()
L = hiplib.lib()
st = (C.c_ulonglong * 64)(); L.agt_debug_lk_stamps_step(st)
t0 = st[0]
f = lambda i: (st[i] - t0) / 2100.0
print("last frame of corner 0: loads issued %.2f us, tiles in LDS %.2f us, end %.2f us" % (f(1), f(2), f(3)))
for lv in (2, 1, 0):
    b = 8 + lv * 8
    print("   level %d: start %.2f  scharr+%.2f  patch+sums+%.2f  iter0+%.2f  iters(n=%d)+%.2f" % (
        lv, f(b), f(b + 1) - f(b), f(b + 2) - f(b + 1), f(b + 3) - f(b + 2), st[b + 6], f(b + 4) - f(b + 2)))
mx = [st[40 + i] / 2100.0 for i in range(16)]
print("slowest corner per frame index (us):", " ".join("%.1f" % v for v in mx))
print("mean corner-frame %.2f us over %d corner-frames; frames in the general loop: %d" % (st[56] / max(st[57], 1) / 2100.0, st[57], st[58]))
