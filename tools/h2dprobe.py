#!/usr/bin/env python3
"""Does hipMemcpyAsync (agt_upload) from torch-pinned memory return before the copy is done?  Host return time vs total."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from accurate_aprilgroup_tracking_amd import cv_hip, hiplib as HL
n = 8 * 1280 * 720
host = torch.empty(8 * n, dtype=torch.uint8).pin_memory()
dev = torch.empty(8 * n, dtype=torch.uint8, device="cuda")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    ctx = cv_hip.Context(64, 64, max_level=0)
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(8):
        HL.check(ctx.L.agt_upload(ctx.h, C.c_void_p(dev.data_ptr() + k * n), C.c_void_p(host.data_ptr() + k * n), n), "up")
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("8 uploads of %.1f MB: host returned after %.0f us, done after %.0f us" % (n / 1e6, (t1 - t0) * 1e6, (t2 - t0) * 1e6))
# event + wait costs
e = torch.cuda.Event(); m = torch.cuda.current_stream()
t0 = time.perf_counter()
for k in range(1000):
    e.record(s); m.wait_event(e)
t1 = time.perf_counter()
print("event record + stream wait: %.1f us per pair" % ((t1 - t0) * 1e3))
