import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from accurate_aprilgroup_tracking_amd import hiplib
hiplib.LIB_PATH = os.path.join(os.path.dirname(hiplib.LIB_PATH), os.environ.get("AGT_LIB", "libagt_hip_knobs.so"))
from accurate_aprilgroup_tracking_amd import synthetic as syn
from accurate_aprilgroup_tracking_amd.tracker import StreamTracker
W, H = 1280, 720
seq = syn.Sequence(W, H, n_frames=8, seed=0, supersample=2)
fr = torch.from_numpy(seq.frames()).cuda()
B = 1
ring = torch.stack([fr[(i % 8) if (i // 8) % 2 == 0 else 7 - (i % 8)].unsqueeze(0).expand(B, H, W) for i in range(32)]).contiguous()
trk = StreamTracker(W, H, seq.obj, seq.K, None, n_streams=B)
c0 = torch.from_numpy(np.repeat(seq.corners(0)[None], B, 0)).cuda().contiguous()
trk.pipeline(int(os.environ.get("AGT_DEPTH", "1")))
trk.reset(ring[0], c0)
for k in range(20): trk.step(ring[(k + 1) % 32])
torch.cuda.synchronize()
K = 416
t0 = time.perf_counter()
if os.environ.get("AGT_CLIP"):          # frames handed over as 32-frame clips (no Python per step)
    for k in range(0, K, 32): trk.step_many(ring)
else:
    for k in range(K): trk.step(ring[(k + 21) % 32])
trk.join(); torch.cuda.synchronize()
t2 = time.perf_counter()
print("depth=%s skip=%s total %.2f us/step" % (os.environ.get("AGT_DEPTH", "1"), os.environ.get("AGT_STEP_SKIP", "0"), (t2 - t0) / K * 1e6))
