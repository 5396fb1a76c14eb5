"""The build's resource claims as a test: DESIGN.md says no kernel of libagt_hip.so has a private segment (scratch)
-- achieved with -mllvm -disable-machine-licm on ONE translation unit (agt_step_nolicm.hip), per-frame opaque solver inputs and forced
inlining -- and that every kernel fits the 512-register file.  The code objects of the built library say whether that still
holds; a compiler or source change that brings scratch back, or drops / adds a kernel instantiation, fails here on the CPU.
(Checked by hand with round 6's sources: agt_step_nolicm.o built WITHOUT its switch gives pnp_group_coop_kernel 512 registers, 250 spilled
VGPRs and 44 B of scratch -- test_no_kernel_has_scratch fails; profiles/r06_kernel_resources.txt holds the figures of the shipped build.)"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
LIB = os.path.join(ROOT, "accurate_aprilgroup_tracking_amd", "libagt_hip.so")


@pytest.fixture(scope="module")
def rows():
    import kernel_resources
    if not os.path.exists(LIB):
        import __graft_entry__
        __graft_entry__.build()
    r = kernel_resources.kernel_rows(LIB)
    assert r, "no gfx950 code object found in %s" % LIB
    return r


def test_no_kernel_has_scratch(rows):
    bad = [(r["name"], r["scratch"]) for r in rows if r["scratch"] != 0]
    assert not bad, "kernels with a private segment (scratch): %s" % bad


def test_every_kernel_fits_the_register_file(rows):
    # .vgpr_count is architectural + accumulation registers of the unified file: 512 per lane on gfx950; 106 SGPRs is the hardware's limit
    bad = [(r["name"], r["vgpr"], r["sgpr"]) for r in rows if r["vgpr"] > 512 or r["sgpr"] > 106]
    assert not bad, bad
    # the one-wave LK kernel of big batches must leave room for four waves per SIMD (128 registers): configs[2]'s throughput rests on it
    lk = [r for r in rows if "lk_kernel<21, 1, 3, 4>" in r["name"]]
    assert len(lk) == 1 and lk[0]["vgpr"] <= 128 and lk[0]["vspill"] == 0, lk


def test_kernel_list_is_the_committed_one(rows):
    want = [l.rstrip("\n") for l in open(os.path.join(ROOT, "tests", "golden", "kernel_list.txt")) if l.strip()]
    got = [r["name"] for r in rows]
    assert sorted(got) == sorted(want), "kernels missing: %s; not in the list: %s" % (sorted(set(want) - set(got)), sorted(set(got) - set(want)))
