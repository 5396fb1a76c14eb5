#!/usr/bin/env python3
"""Disassembly of one kernel of a built library: python tools/isa_dump.py '<substring of the demangled name>' [library] > out.s
(development aid; the code objects are unbundled into a temporary directory)"""
import os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa_mix import code_objects, LLVM, ROOT


def main():
    want = sys.argv[1]
    lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "accurate_aprilgroup_tracking_amd", "libagt_hip.so")
    with tempfile.TemporaryDirectory() as tmp:
        for co in code_objects(lib, tmp):
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
            dem = subprocess.run(["c++filt"], input=dis, capture_output=True, text=True).stdout
            lines = dem.splitlines()
            starts = [i for i, l in enumerate(lines) if re.match(r"^[0-9a-f]+ <.*>:$", l)]
            for si, i in enumerate(starts):
                if want in lines[i]:
                    print("\n".join(lines[i:(starts[si + 1] if si + 1 < len(starts) else len(lines))]))
                    return
    print("kernel not found:", want, file=sys.stderr)


if __name__ == "__main__":
    main()
