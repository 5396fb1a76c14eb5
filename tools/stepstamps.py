#!/usr/bin/env python3
"""Role timeline of the fused single-stream step (diagnostic library, make -C csrc dbg): entry / exit of the PnP block, of
the first LK block, latest exit of the LK and pyramid roles, relative to the earliest block entry; averaged over N steps."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from accurate_aprilgroup_tracking_amd import hiplib
hiplib.LIB_PATH = os.path.join(os.path.dirname(hiplib.LIB_PATH), os.environ.get("AGT_LIB", "libagt_hip_dbg.so"))
from accurate_aprilgroup_tracking_amd import synthetic as syn
from accurate_aprilgroup_tracking_amd.tracker import StreamTracker
W, H = 1280, 720
seq = syn.Sequence(W, H, n_frames=12, seed=0, supersample=2)
fr = torch.from_numpy(seq.frames()).cuda()
order = list(range(1, 12)) + list(range(10, -1, -1))
trk = StreamTracker(W, H, seq.obj, seq.K, None, n_streams=1)
D = int(sys.argv[1]) if len(sys.argv) > 1 else 1      # frames per launch
trk.pipeline(D)
trk.reset(fr[0:1].contiguous(), torch.from_numpy(seq.corners(0)[None]).cuda().contiguous())
L = hiplib.lib()
L.agt_debug_step_stamps.argtypes = [C.c_void_p, C.c_int]
st = (C.c_ulonglong * 16)()
acc = np.zeros(6); n = 0
role = np.zeros(2)
for i in range(40):
    torch.cuda.synchronize(); L.agt_debug_step_stamps(st, 1)
    for j in range(D):
        k = order[(i * D + j) % len(order)]
        trk.step(fr[k:k + 1])
    torch.cuda.synchronize(); L.agt_debug_step_stamps(st, 0)
    if i < 10: continue
    t0 = st[6]
    acc += np.array([st[0] - t0, st[1] - t0, st[2] - t0, st[3] - t0, st[4] - t0, st[5] - t0], float); n += 1
    role += np.array([st[1] - st[0], st[3] - st[2]], float)
acc /= n * 2100.0; role /= n * 2100.0 * D
print("depth %d, us after the first block entry (2.1 GHz assumed; blocks on different XCDs have different counters): PnP block in %.2f out %.2f | LK block 0 in %.2f out %.2f | last LK out %.2f | last pyramid out %.2f"
      % ((D,) + tuple(acc)))
print("per frame (own counter, exit - entry over %d frames): PnP role %.2f us, LK role (block 0) %.2f us" % (D, role[0], role[1]))
