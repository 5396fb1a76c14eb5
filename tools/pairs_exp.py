#!/usr/bin/env python3
"""What bounds the pipelined cold-pair step (bench.py --workload c3pairs)?  Ablations of the SAME bench flow, chosen by PAIRS_EXP
(development aid, GPU box only; the lines it prints are NOT measurements of the workload -- a stage is missing or shortened):
    nolk      agt_lk_track not launched            lk1      LK with max_count 1 (its VALU work cut to the fixed part)
    nopnp     agt_solve_pnp not launched           nonext   the second pyramid of every pair not built
    nopyr     no pyramid built at all
AGT_LIB selects the library (e.g. libagt_hip_knobs.so for AGT_PYR4 / AGT_PYR5 knobs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from accurate_aprilgroup_tracking_amd import hiplib
if os.environ.get("AGT_LIB"):
    hiplib.LIB_PATH = os.path.join(os.path.dirname(hiplib.LIB_PATH), os.environ["AGT_LIB"])
L = hiplib.lib()
exp = os.environ.get("PAIRS_EXP", "")


class Wrap:
    """the ctypes library with some entry points replaced (bench_pairs calls through ctx.L)"""
    def __init__(self, lib):
        object.__setattr__(self, "_lib", lib)
        self.n = 0

    def __getattr__(self, name):
        f = getattr(self._lib, name)
        if name == "agt_lk_track" and exp == "nolk":
            return lambda *a: 0
        if name == "agt_lk_track" and exp == "lk1":
            return lambda *a: f(*(a[:10] + (1,) + a[11:]))
        if name == "agt_solve_pnp" and exp == "nopnp":
            return lambda *a: 0
        if name == "agt_pyramid_build" and exp == "nopyr":
            return lambda *a: 0
        if name == "agt_pyramid_build" and exp == "nonext":
            return lambda *a: (0 if a[1] == 1 and self.warm() else f(*a))
        return f

    def warm(self):
        self.n += 1
        return self.n > 8          # the first calls build slot 1 once, so that LK has valid levels to read


from accurate_aprilgroup_tracking_amd import cv_hip
_orig = cv_hip.Context.__init__


def _init(self, *a, **k):
    _orig(self, *a, **k)
    self.L = Wrap(self.L)


cv_hip.Context.__init__ = _init
sys.argv = ["bench.py"] + sys.argv[1:]
import bench
bench.main()
