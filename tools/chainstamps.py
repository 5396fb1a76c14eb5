#!/usr/bin/env python3
"""Timeline of corner 0 of the frame-chained LK role (agt_lk_chain_body.h) over the first four frames of a 16-frame launch,
from in-kernel s_memtime stamps (diagnostic library: make -C csrc dbg)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from accurate_aprilgroup_tracking_amd import hiplib
hiplib.LIB_PATH = os.path.join(os.path.dirname(hiplib.LIB_PATH), os.environ.get("AGT_LIB", "libagt_hip_dbg.so"))
from accurate_aprilgroup_tracking_amd import synthetic as syn
from accurate_aprilgroup_tracking_amd.tracker import StreamTracker
W, H, D = 1280, 720, 16
seq = syn.Sequence(W, H, n_frames=24, seed=0, supersample=2)
fr = torch.from_numpy(seq.frames()).cuda()
trk = StreamTracker(W, H, seq.obj, seq.K, None, n_streams=1)
trk.pipeline(D)
trk.reset(fr[0:1].contiguous(), torch.from_numpy(seq.corners(0)[None]).cuda().contiguous())
L = hiplib.lib()
L.agt_debug_chain_stamps.argtypes = [C.c_void_p]
clip = fr[1:17].unsqueeze(1).contiguous()
trk.step_many(clip); trk.join(); torch.cuda.synchronize()
st = (C.c_ulonglong * 64)(); L.agt_debug_chain_stamps(st)
GHZ = float(os.environ.get("AGT_GHZ", "2.33"))
for k in range(4):
    s = [st[k * 16 + i] for i in range(16)]
    f = lambda i: (s[i] - s[0]) / (GHZ * 1e3)
    it = s[14]
    print("   geometry done %.2f, counters done %.2f" % (f(7), f(15)))
    print("frame %d (us at %.2f GHz): loads issued %.2f, tiles ready %.2f, scharr %.2f, patch+sums %.2f | L2 %.2f..%.2f (%d it) L1 %.2f..%.2f (%d it) L0 %.2f..%.2f (%d it) | published %.2f, end %.2f"
          % (k, GHZ, f(1) if s[1] > s[0] else 0.0, f(2), f(3), f(4), f(12), f(13), (it >> 16) & 255, f(10), f(11), (it >> 8) & 255, f(8), f(9), it & 255, f(5), f(6)))
    if k: print("   frame start %.2f us after the previous frame's start" % ((s[0] - prev0) / (GHZ * 1e3)))
    prev0 = s[0]
