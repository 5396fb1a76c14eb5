#!/usr/bin/env python3
"""Per-frame timeline of the PnP role (stream 0) inside one chained launch of the fused step (diagnostic library: make -C csrc dbg):
loop top -> corners loaded and counted (before the wait for the previous frame's state) -> state acquired -> frame done."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from accurate_aprilgroup_tracking_amd import hiplib
hiplib.LIB_PATH = os.path.join(os.path.dirname(hiplib.LIB_PATH), os.environ.get("AGT_LIB", "libagt_hip_dbg.so"))
from accurate_aprilgroup_tracking_amd import synthetic as syn
from accurate_aprilgroup_tracking_amd.tracker import StreamTracker
W, H = 1280, 720
D = int(sys.argv[1]) if len(sys.argv) > 1 else 16
seq = syn.Sequence(W, H, n_frames=12, seed=0, supersample=2)
fr = torch.from_numpy(seq.frames()).cuda()
order = list(range(1, 12)) + list(range(10, -1, -1))
trk = StreamTracker(W, H, seq.obj, seq.K, None, n_streams=1)
trk.pipeline(D)
trk.reset(fr[0:1].contiguous(), torch.from_numpy(seq.corners(0)[None]).cuda().contiguous())
L = hiplib.lib()
L.agt_debug_role_stamps.argtypes = [C.c_void_p]
for i in range(4 * D):
    trk.step(fr[order[i % len(order)]:order[i % len(order)] + 1])
torch.cuda.synchronize()
st = (C.c_ulonglong * 128)(); L.agt_debug_role_stamps(st)
TICK = float(os.environ.get("AGT_TICK_NS", "10.0"))        # s_memtime: 100 MHz on gfx950
s = np.array(st[:], np.float64).reshape(32, 4) * TICK / 1e3
t0 = s[0, 0]
for k in range(D):
    print("frame %2d: top %7.2f | corners +%5.2f | state +%5.2f (waited %5.2f) | done +%5.2f | period %5.2f" % (
        k, s[k, 0] - t0, s[k, 1] - s[k, 0], s[k, 2] - s[k, 0], s[k, 2] - s[k, 1], s[k, 3] - s[k, 2], (s[k, 3] - s[k - 1, 3]) if k else 0.0))
