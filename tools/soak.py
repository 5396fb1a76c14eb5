#!/usr/bin/env python3
"""Soak of the SHIPPED library on the 64-stream split pipeline (round 6, after VERDICT r5 #1): block after block of the same clip from the same
reset state; every block's records must be BITWISE those of the first block, the copies of a seed (stream b shows seed b % 4) bitwise equal to
one another, no record flagged AGT_TRK_CHAIN_TIMEOUT, agt_synchronize clean.  The pipeline is deterministic (exact integer LK sums, fixed
reduction orders), so a pointer read too early, a stale table entry or a lost update -- the class of defect the max-ILP object of round 5 had
once per ~1,000 launches, silently -- shows as a differing block.
    python tools/soak.py [seconds] [streams] [steps per block]        (default 240 s, 64 streams, 256 steps)
    python tools/soak.py [seconds] dense [steps per block]            (BASELINE configs[4]: 60 tags / 240 corners + the dense stage, one stream, clips: the
                                                                       chained LK | four-wave PnP launch and the Gauss-Newton launches; pose AND dense records)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from accurate_aprilgroup_tracking_amd import synthetic as syn, hiplib as H
from accurate_aprilgroup_tracking_amd.tracker import StreamTracker

SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
DENSE = len(sys.argv) > 2 and sys.argv[2] == "dense"
B = 1 if DENSE else (int(sys.argv[2]) if len(sys.argv) > 2 else 64)
K = int(sys.argv[3]) if len(sys.argv) > 3 else 256
W, Hh, NF = 1280, 720, 8
seqs = ([syn.Sequence(W, Hh, n_tags=60, n_frames=NF, seed=0, supersample=3, group_seed=0)] * 4 if DENSE else
        [syn.Sequence(W, Hh, n_frames=NF, seed=s, supersample=2, group_seed=0) for s in range(4)])
ring = torch.from_numpy(np.stack([np.stack([seqs[b % 4].frame(k) for b in range(B)]) for k in range(NF)])).cuda().contiguous()   # [NF, B, H, W]
order = [(i % NF) if (i // NF) % 2 == 0 else NF - 1 - (i % NF) for i in range(2 * NF)]           # ping-pong over the rendered frames
trk = StreamTracker(W, Hh, seqs[0].obj, seqs[0].K, None, n_streams=B)
trk.pipeline(0 if DENSE else 16)
dn = None
if DENSE:
    mx = syn.model_samples(seqs[0].group, 32)
    T = np.nan_to_num(syn.sample_bilinear(seqs[0].frame(0), syn.project(mx, seqs[0].rvecs[0], seqs[0].tvecs[0], seqs[0].K)), nan=128.0).astype(np.float32)
    trk.dense_model(torch.from_numpy(mx).cuda(), torch.from_numpy(T).cuda(), iters=5, photo_weight=0.05, reseed=True)
    dn = torch.zeros((K, B, H.DENSE_STRIDE), dtype=torch.float64, device="cuda")
    clip_idx = [order[(k + 1) % len(order)] for k in range(K)]
    clip = ring[clip_idx].contiguous()                     # [K, 1, H, W]: the block's frames as one clip
c0 = torch.from_numpy(np.stack([seqs[b % 4].corners(0) for b in range(B)])).cuda().contiguous()
so = torch.zeros((K, B, H.STATE_STRIDE), dtype=torch.float64, device="cuda")

def block():
    so.zero_()
    trk.reset(ring[0], c0)
    if DENSE:
        dn.zero_()
        trk.step_many_dense(clip, so, dn)
    else:
        for k in range(K):
            trk.step(ring[order[(k + 1) % len(order)]], so[k])
    trk.join()
    rc = trk.ctx.L.agt_synchronize(trk.ctx.h)
    rec = so.cpu().numpy()
    return rc, (np.concatenate([rec, dn.cpu().numpy()], axis=2) if DENSE else rec)

rc, first = block()
assert rc == 0, "agt_synchronize %d" % rc
ok0 = float(first[:, :, H.ST_OK].mean())
print("block 0: accepted %.4f, LM iterations %.2f" % (ok0, float(first[:, :, H.ST_ITERS].mean())), flush=True)
ref = first.view(np.uint64)
bad_blocks = copies_bad = flagged = 0
n = 1
t0 = time.time(); t_print = t0
while time.time() - t0 < SECONDS:
    rc, rec = block()
    n += 1
    r = rec.view(np.uint64)
    if rc != 0 or not np.array_equal(r, ref):
        bad_blocks += 1
        d = np.argwhere(r != ref)
        print("block %d differs (agt_synchronize %d): first at frame %d stream %d word %d, %d words" % (n - 1, rc, *(d[0] if len(d) else (-1, -1, -1)), len(d)), flush=True)
    if not all(np.array_equal(r[:, q], r[:, q % 4]) for q in range(B)):
        copies_bad += 1
    flagged += int(((rec[:, :, 11].astype(np.int64) & 512) != 0).sum())
    if time.time() - t_print > 30:
        t_print = time.time()
        print("  %6.0f s: %d blocks, %d differing, %d with unequal copies, %d flagged records" % (t_print - t0, n, bad_blocks, copies_bad, flagged), flush=True)
dt = time.time() - t0
fused = B * c0.shape[1] <= 256 and not DENSE              # (agt_step_fits: the fused chained step; else the split pipeline)
groups = n * (K // 16 + 4)                  # launch groups per block: 16 frames each, plus the fill / drain ramp
what = ("%d frames of 6 launches: LK(240) | PnP(240) chained + 5 Gauss-Newton launches (dense clips)" % (n * K)) if DENSE else ("~%d chained LK | PnP launches (fused step)" % groups) if fused else ("~%d group pose launches, ~%d per-frame LK launches (split pipeline)" % (groups, 2 * n * K))
print("soak: %d blocks of %d steps x %d streams in %.0f s (%.1f M stream-frames, %s): %d blocks differ from block 0, "
      "%d blocks with unequal copies, %d flagged records, accepted %.4f"
      % (n, K, B, dt, n * K * B / 1e6, what, bad_blocks, copies_bad, flagged, ok0), flush=True)
sys.exit(1 if (bad_blocks or copies_bad or flagged) else 0)
