import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from accurate_aprilgroup_tracking_amd import synthetic as syn
from accurate_aprilgroup_tracking_amd.tracker import StreamTracker
W, H = 1280, 720
seq = syn.Sequence(W, H, n_frames=8, seed=0, supersample=2)
fr = torch.from_numpy(seq.frames()).cuda()
ring = torch.stack([fr[(i % 8) if (i // 8) % 2 == 0 else 7 - (i % 8)].unsqueeze(0) for i in range(32)]).contiguous()
trk = StreamTracker(W, H, seq.obj, seq.K, None, n_streams=1)
trk.pipeline(int(os.environ.get("AGT_DEPTH", "4")))
c0 = torch.from_numpy(seq.corners(0)[None]).cuda().contiguous()
for K in [int(x) for x in os.environ.get("AGT_KS", "400,1600,2000,3000").split(",")]:
    trk.reset(ring[0], c0)
    so = torch.zeros((K, 1, 16), dtype=torch.float64, device="cuda") if os.environ.get("WITH_SO") else None
    for k in range(40): trk.step(ring[(k + 1) % 32])
    trk.join(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks = []
    for k in range(K):
        trk.step(ring[(k + 41) % 32], so[k] if so is not None else None)
        if (k + 1) % max(400, K // 8) == 0: marks.append(time.perf_counter() - t0)
    t1 = time.perf_counter()
    trk.join(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("K=%d enqueue %.1f ms (marks %s) total %.1f ms -> %.2f us/step" % (K, (t1 - t0) * 1e3, ["%.1f" % (m * 1e3) for m in marks], (t2 - t0) * 1e3, (t2 - t0) / K * 1e6))
