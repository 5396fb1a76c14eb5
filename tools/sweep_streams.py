#!/usr/bin/env python3
"""Stream-count sweep of the per-frame step: B = 1 .. 128 streams of 1280x720 / 48 corners, frames/s and us per
step, with the launch form the library picks ("auto") and with the fused / split form forced.

Forcing needs the knobs build (make -C accurate_aprilgroup_tracking_amd/csrc knobs: AGT_STEP_MAX_CORNERS is read from the
environment there; the product library reads no environment).  One child process per point (the knob is read once).

    python tools/sweep_streams.py [--streams 1,2,4,...] [--modes auto,fused,split] > profiles/r03_stream_sweep.txt
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys, json
sys.path.insert(0, os.environ["AGT_ROOT"])
from accurate_aprilgroup_tracking_amd import hiplib
if os.environ.get("AGT_LIB"):
    hiplib.LIB_PATH = os.environ["AGT_LIB"]
sys.argv = ["bench.py"] + json.loads(os.environ["AGT_ARGV"])
import bench
bench.main()
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", default="1,2,4,8,16,32,42,43,64,128")
    ap.add_argument("--modes", default="auto,fused,split")
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--depth", type=int, default=16)
    args = ap.parse_args()
    knobs = os.path.join(ROOT, "accurate_aprilgroup_tracking_amd", "libagt_hip_knobs.so")
    print("# streams mode frames_per_s us_per_step us_per_step_per_stream_x64 accepted   (1280x720, 48 corners, %d frames per launch group, %d steps per block, median of 7)"
          % (args.depth, args.steps))
    for B in [int(x) for x in args.streams.split(",")]:
        for mode in args.modes.split(","):
            env = dict(os.environ, AGT_ROOT=ROOT)
            if mode != "auto":
                if not os.path.exists(knobs):
                    print("# %s needs %s (make knobs)" % (mode, knobs)); continue
                env["AGT_LIB"] = knobs
                env["AGT_STEP_MAX_CORNERS"] = "1000000000" if mode == "fused" else "0"
            if mode == "fused" and B * 48 > 128 * 48:
                continue
            env["AGT_ARGV"] = json.dumps(["--workload", "c2", "--streams", str(B), "--steps", str(args.steps), "--warmup", "16", "--depth", str(args.depth),
                                          "--blocks", "7", "--no-extras", "--no-cpu-baseline", "--render-frames", "8"])
            r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode or not line:
                print("%d %s FAILED rc=%d %s" % (B, mode, r.returncode, r.stderr[-300:].replace("\n", " | "))); continue
            d = json.loads(line[-1])
            us = d["ms_per_step"] * 1e3
            print("%4d %-5s %12.1f %9.2f %9.2f %6.3f" % (B, mode, d["value"], us, us / B * 64, d["accepted_frac"]), flush=True)


if __name__ == "__main__":
    main()
