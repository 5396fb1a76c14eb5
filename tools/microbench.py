#!/usr/bin/env python3
"""Host-overhead and batch-scaling probe for the fused step (development aid, GPU box only)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from accurate_aprilgroup_tracking_amd import synthetic as syn, hiplib as HL
from accurate_aprilgroup_tracking_amd.tracker import StreamTracker

W, H = 1280, 720
seq = syn.Sequence(W, H, n_frames=8, seed=0, supersample=2)
fr = torch.from_numpy(seq.frames()).cuda()
for B, pipe in [(b, p) for b in (1, 2, 4, 8, 16, 32, 64, 128) for p in (0, 1)]:
    ring = torch.stack([fr[(i % 8) if (i // 8) % 2 == 0 else 7 - (i % 8)].unsqueeze(0).expand(B, H, W) for i in range(32)]).contiguous()
    trk = StreamTracker(W, H, seq.obj, seq.K, None, n_streams=B)
    trk.pipeline(pipe)
    c0 = torch.from_numpy(np.repeat(seq.corners(0)[None], B, 0)).cuda().contiguous()
    trk.reset(ring[0], c0)
    for k in range(20):
        trk.step(ring[(k + 1) % 32])
    torch.cuda.synchronize()
    K = 200
    t0 = time.perf_counter()
    for k in range(K):
        trk.step(ring[(k + 21) % 32])
    t1 = time.perf_counter()
    trk.join(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("B=%d pipe=%d enqueue %.1f us/step, total %.1f us/step, %.0f frames/s" % (B, pipe, (t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6, B * K / (t2 - t0)))
    continue
    # raw ctypes call cost with prebuilt pointers
    L = trk.ctx.L; h = trk.ctx.h
    ptrs = [ring[i].data_ptr() for i in range(32)]
    pitch, bs = ring.stride(2), ring.stride(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        L.agt_track_frame(h, ptrs[(k + 1) % 32], pitch, bs, B, None)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("   raw ctypes: enqueue %.1f us/step, total %.1f us/step, %.0f frames/s" % ((t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6, B * K / (t2 - t0)))
