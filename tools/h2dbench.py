#!/usr/bin/env python3
"""bench.py's H2D-inclusive side measurement alone (for rocprofv3 --kernel-trace --memory-copy-trace timelines)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
args = argparse.Namespace(steps=200, warmup=10, streams=None, render_frames=24, dry_run=False, depth=1, blocks=3)
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
b = B.Bench(torch, B.WORKLOADS["c2"], args, 0, 1, dev)
print(B.h2d_inclusive(torch, b, 200))
