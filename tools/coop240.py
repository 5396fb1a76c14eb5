#!/usr/bin/env python3
"""Split-mode pose role with 64 < n <= 256 corners (the four-wave solver): us per step of B streams of 240 corners at pipeline
depth D, for the A/B of VERDICT r4 #4 -- per-frame pnp_coop_kernel launches (shipped, no scratch) against the group launch
pnp_group_coop_kernel (knobs build, AGT_PNP_COOP_GROUP=1: 129 VGPR spills / 376 B of scratch).  AGT_LIB selects the library."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from accurate_aprilgroup_tracking_amd import hiplib
if os.environ.get("AGT_LIB"):
    hiplib.LIB_PATH = os.path.join(os.path.dirname(hiplib.LIB_PATH), os.environ["AGT_LIB"])
from accurate_aprilgroup_tracking_amd import synthetic as syn
from accurate_aprilgroup_tracking_amd.tracker import StreamTracker
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
D = int(sys.argv[2]) if len(sys.argv) > 2 else 8
K = 64
s = syn.Sequence(1280, 720, n_tags=60, n_frames=6, seed=8, supersample=2)
fr = torch.from_numpy(s.frames()).cuda()
trk = StreamTracker(1280, 720, s.obj, s.K, None, n_streams=B)
trk.pipeline(D)
order = [1, 2, 3, 4, 5, 4, 3, 2, 1, 0]
clip = torch.stack([fr[order[i % len(order)]].unsqueeze(0).expand(B, -1, -1) for i in range(K)]).contiguous()
so = trk.new_state_buffer(K)
def run():
    trk.reset(fr[0:1].expand(B, -1, -1).contiguous(), torch.from_numpy(np.stack([s.corners(0)] * B)).cuda().contiguous())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    trk.step_many(clip, so); trk.join(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e6
run(); run()
ts = sorted(run() for _ in range(9))
st = so.cpu().numpy()
print("B=%d depth=%d 240 corners: %.2f us/step (median of 9; min %.2f), accepted %.3f, checksum %.12g" % (B, D, ts[4], ts[0], st[:, :, hiplib.ST_OK].mean(), float(np.abs(st[:, :, :6]).sum())))
