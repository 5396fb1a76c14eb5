"""where the host time of a 20-step block goes: run(20) | join | sync (knobs library via AGT_LIB)"""
import os, sys, time, argparse
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from accurate_aprilgroup_tracking_amd import hiplib
if os.environ.get("AGT_LIB"):
    hiplib.LIB_PATH = os.path.join(os.path.dirname(hiplib.LIB_PATH), os.environ["AGT_LIB"])
import torch, bench
K = int(os.environ.get("K", "20")); DEPTH = int(os.environ.get("DEPTH", str(K)))
args = argparse.Namespace(steps=K, warmup=5, render_frames=24, streams=None, dry_run=False, per_step_calls=False)
b = bench.Bench(torch, bench.WORKLOADS["c2"], args, 0, 1, torch.device("cuda", 0))
b.trk.pipeline(DEPTH)
state = torch.zeros((K, b.B, 16), dtype=torch.float64, device="cuda")
b.restart(); b.run(5, None); b.trk.join(); torch.cuda.synchronize()
T = []
for r in range(60):
    if b.since + K > bench.REDETECT: b.refresh()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); b.run(K, state); t1 = time.perf_counter(); b.trk.join(); t2 = time.perf_counter(); torch.cuda.synchronize(); t3 = time.perf_counter()
    if r >= 10: T.append((t1 - t0, t2 - t1, t3 - t2, t3 - t0))
T = np.median(np.array(T) * 1e6, axis=0)
print("K=%d depth=%d: run %.1f | join %.1f | sync %.1f | total %.1f us -> %.1f k frames/s" % (K, DEPTH, T[0], T[1], T[2], T[3], K / T[3] * 1e3))
