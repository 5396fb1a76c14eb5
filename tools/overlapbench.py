#!/usr/bin/env python3
"""Do the pyramid, LK and PnP kernels of a 64-stream step overlap when they run on different HIP streams?
(development aid, GPU box only): time each alone and all three concurrently on independent data."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from accurate_aprilgroup_tracking_amd import synthetic as syn, cv_hip
W, H, B = 1280, 720, 64
seq = syn.Sequence(W, H, n_frames=2, seed=0, supersample=2)
f0 = torch.from_numpy(seq.frame(0)).cuda().unsqueeze(0).expand(B, H, W).contiguous()
f1 = torch.from_numpy(seq.frame(1)).cuda().unsqueeze(0).expand(B, H, W).contiguous()
pts = torch.from_numpy(np.repeat(seq.corners(0)[None], B, 0)).cuda().contiguous()
obj = torch.from_numpy(seq.obj.astype(np.float32)).cuda()
guess = np.concatenate([seq.rvecs[0], seq.tvecs[0]])
sA, sB, sC = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
ctxs = {}
for name, st in (("pyr", sA), ("lk", sB), ("pnp", sC)):
    with torch.cuda.stream(st):
        c = cv_hip.Context(W, H, max_level=2, max_points=48, max_streams=B)
        c.pyramid_build(0, f0); c.pyramid_build(1, f1)
        ctxs[name] = c
with torch.cuda.stream(sB):
    nx, stt, er = ctxs["lk"].lk_track(0, 1, pts)
img = nx.clone()
pose0 = torch.from_numpy(np.repeat(guess[None], B, 0)).cuda().contiguous()
pose = pose0.clone()
torch.cuda.synchronize()

def run(which, n):
    for _ in range(n):
        if "pyr" in which:
            with torch.cuda.stream(sA):
                ctxs["pyr"].use_current_stream(); ctxs["pyr"].pyramid_build(0, f0)
        if "lk" in which:
            with torch.cuda.stream(sB):
                ctxs["lk"].use_current_stream(); ctxs["lk"].lk_track(0, 1, pts, nx, want_err=False)
        if "pnp" in which:
            with torch.cuda.stream(sC):
                ctxs["pnp"].use_current_stream(); ctxs["pnp"].solve_pnp(obj, img, seq.K, None, pose, True)

for which in (("pyr",), ("lk",), ("pnp",), ("pyr", "lk"), ("pyr", "lk", "pnp")):
    run(which, 10); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(which, 200); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("%-16s %7.1f us per round" % ("+".join(which), (t1 - t0) / 200 * 1e6))
