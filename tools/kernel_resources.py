#!/usr/bin/env python3
"""Register / spill / scratch / LDS figures of every kernel in a built library, from the code-object notes
(llvm-readelf --notes on the gfx950 code objects bundled in the .so).  VERDICT r2 #3: spill counts of the shipped kernels.

    python tools/kernel_resources.py [accurate_aprilgroup_tracking_amd/libagt_hip.so] > profiles/r03_kernel_resources.txt
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, out))


def kernel_rows(lib):
    """-> one dict per kernel of the library's gfx950 code objects: name (demangled), vgpr (arch + acc), agpr, sgpr, vspill,
    sspill, scratch (private segment bytes), lds (static), wg -- integers (tests/test_kernel_resources.py holds the product
    library to them)"""
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        # the fat binary section holds one offload bundle per translation unit
        sec = os.path.join(tmp, "fatbin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + sec, lib, os.path.join(tmp, "x")], stderr=subprocess.DEVNULL)
        blob = open(sec, "rb").read()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)] + [len(blob)]
        for i in range(len(starts) - 1):
            b = os.path.join(tmp, "b%d" % i)
            open(b, "wb").write(blob[starts[i]:starts[i + 1]])
            co = os.path.join(tmp, "co%d" % i)
            r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + b, "--output=" + co,
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], capture_output=True, text=True)
            if r.returncode or not os.path.exists(co):
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
            for blk in notes.split("- .agpr_count:")[1:]:
                f = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
                rows.append(dict(name=f("name"), vgpr=f("vgpr_count"), agpr=blk.split()[0], sgpr=f("sgpr_count"), vspill=f("vgpr_spill_count"),
                                 sspill=f("sgpr_spill_count"), scratch=f("private_segment_fixed_size"), lds=f("group_segment_fixed_size"),
                                 wg=f("max_flat_workgroup_size")))
    dm = demangle([r["name"] for r in rows])
    for r in rows:
        r["name"] = dm[r["name"]]
        for k in ("vgpr", "agpr", "sgpr", "vspill", "sspill", "scratch", "lds", "wg"):
            r[k] = int(r[k])
    return sorted(rows, key=lambda r: r["name"])


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "accurate_aprilgroup_tracking_amd", "libagt_hip.so")
    rows = kernel_rows(lib)
    dm = {r["name"]: r["name"] for r in rows}
    print("# %s" % os.path.relpath(lib, ROOT))
    print("# vgpr agpr sgpr vgpr_spill sgpr_spill scratch_B static_lds_B max_wg  kernel")
    for r in rows:
        print("%4s %4s %4s %6s %6s %7s %7s %5s  %s" % (r["vgpr"], r["agpr"], r["sgpr"], r["vspill"], r["sspill"], r["scratch"], r["lds"], r["wg"], dm[r["name"]][:150]))


if __name__ == "__main__":
    main()
