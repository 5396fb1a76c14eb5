#!/usr/bin/env python3
"""pyrDown alone on 64 distinct 1280x720 frames per launch, rotating over > 256 MiB of sources
(HBM-cold).  Used under rocprofv3 --pmc for the traffic figures (development aid, GPU box only)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from accurate_aprilgroup_tracking_amd import hiplib
if os.environ.get("AGT_LIB"):       # e.g. the knobs build, to sweep AGT_PYR3_OH / AGT_PYR3
    hiplib.LIB_PATH = os.path.join(os.path.dirname(hiplib.LIB_PATH), os.environ["AGT_LIB"])
from accurate_aprilgroup_tracking_amd import cv_hip
W, H, B, SLOTS = 1280, 720, 64, 6
ctx = cv_hip.Context(W, H, max_level=2, max_points=48, max_streams=B)
src = torch.randint(0, 256, (SLOTS, B, H, W), dtype=torch.uint8, device="cuda")      # 354 MB
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
calib = src[0].clone()          # 58,982,400-byte streaming copy: calibrates FETCH_SIZE / WRITE_SIZE in the same pass
for i in range(10):
    ctx.pyr_down(src[i % SLOTS])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(n):
    ctx.pyr_down(src[i % SLOTS])
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / n
byt = B * (W * H + (W // 2) * (H // 2))
print("pyr_down L0->L1 x%d images: %.2f us/launch, %.1f GB/s algorithmic (%.1f%% of 8 TB/s)" % (B, us, byt / us / 1e3, byt / us / 1e3 / 80))
# the two-level pass (L0 -> L1 -> L2 in one launch: agt_pyramid_build with max_level 2), same rotation
for i in range(10):
    ctx.pyramid_build(1, src[i % SLOTS])
torch.cuda.synchronize()
e0.record()
for i in range(n):
    ctx.pyramid_build(1, src[i % SLOTS])
e1.record(); torch.cuda.synchronize()
us2 = e0.elapsed_time(e1) * 1e3 / n
byt2 = B * (W * H + (W // 2) * (H // 2) + (W // 4) * (H // 4))
print("pyramid L0->L1->L2 x%d images, one pass: %.2f us/launch, %.1f GB/s algorithmic (%.1f%% of 8 TB/s)" % (B, us2, byt2 / us2 / 1e3, byt2 / us2 / 1e3 / 80))
