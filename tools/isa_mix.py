#!/usr/bin/env python3
"""Instruction mix of a kernel of the built library, from its ISA (llvm-objdump of the gfx950 code object): whole kernel
(static counts) and every innermost loop (a backward branch whose body contains no other backward branch target), classified.
VERDICT r2 #2: the instruction-mix table of lk_kernel<21,1,3,4>.

    python tools/isa_mix.py 'lk_kernel<21, 1, 3, 4>' [libagt_hip.so | kernel.s] [--path A-B,C-D,...] > profiles/r03_lk_iteration_isa.md

kernel.s: a listing written by tools/isa_dump.py (an object that is no longer built).  --path: address ranges (hex, inclusive) of the
instructions ONE trip of a loop executes on its common path -- the innermost-loop tables count every instruction between a backward
branch and its target, rare paths included; the path table counts what a trip issues.
"""
import os, re, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CLASSES = [
    ("dot4 / dot2 (bilinear taps)", r"^v_dot[24]"), ("perm / alignbyte (byte packing)", r"^v_(perm_b32|alignbyte|alignbit)"),
    ("DPP / lane swaps (reductions)", r"(_dpp|^v_permlane|^v_mov_b32_dpp)"), ("readlane / readfirstlane", r"^v_read(first)?lane"),
    ("integer multiply / mad", r"^v_(mul_i32_i24|mul_u32_u24|mad_i32_i24|mad_u32_u24|mul_lo|mul_hi|mad_u64|mad_i64)"),
    ("float (weights, 2x2 solve, tests)", r"^v_(pk_)?(add|sub|mul|fma|fmac|mac|rndne|floor|cvt|cmp|rcp|rsq|sqrt|max|min|cndmask|trunc|fract|div).*(f32|f64)|^v_cvt_|^v_rndne|^v_floor"),
    ("integer add / shift / logic / select", r"^v_"),
    ("LDS", r"^ds_"), ("global / buffer / flat memory", r"^(global_|buffer_|flat_|scratch_)"),
    ("scalar ALU / moves", r"^s_(?!waitcnt|barrier|cbranch|branch|nop|endpgm|load|buffer_load|sleep|setprio)"),
    ("scalar memory (kernel arguments)", r"^s_(load|buffer_load)"), ("waitcnt", r"^s_waitcnt"), ("branch", r"^s_(c)?branch"),
    ("barrier", r"^s_barrier"), ("nop / other", r"."),
]


def classify(op):
    for name, pat in CLASSES:
        if re.search(pat, op):
            return name
    return "nop / other"


def code_objects(lib, tmp):
    sec = os.path.join(tmp, "fat")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + sec, lib, os.path.join(tmp, "x")], stderr=subprocess.DEVNULL)
    blob = open(sec, "rb").read(); magic = b"__CLANG_OFFLOAD_BUNDLE__"
    st = [m.start() for m in re.finditer(re.escape(magic), blob)] + [len(blob)]
    for i in range(len(st) - 1):
        b = os.path.join(tmp, "b%d" % i); open(b, "wb").write(blob[st[i]:st[i + 1]])
        co = os.path.join(tmp, "co%d" % i)
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + b, "--output=" + co,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], capture_output=True)
        if os.path.exists(co):
            yield co


PATH = []


def main():
    args = sys.argv[1:]
    if "--path" in args:
        k = args.index("--path")
        for r in args[k + 1].split(","):
            a, b = r.split("-"); PATH.append((int(a, 16), int(b, 16)))
        del args[k:k + 2]
    want = args[0]
    lib = args[1] if len(args) > 1 else os.path.join(ROOT, "accurate_aprilgroup_tracking_amd", "libagt_hip.so")
    if lib.endswith(".s"):
        report(want, lib, open(lib).read().splitlines()[1:])
        return
    with tempfile.TemporaryDirectory() as tmp:
        for co in code_objects(lib, tmp):
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
            dem = subprocess.run(["c++filt"], input=dis, capture_output=True, text=True).stdout
            lines = dem.splitlines()
            starts = [i for i, l in enumerate(lines) if re.match(r"^[0-9a-f]+ <.*>:$", l)]
            for si, i in enumerate(starts):
                if want in lines[i]:
                    body = lines[i + 1:(starts[si + 1] if si + 1 < len(starts) else len(lines))]
                    report(want, lib, body)
                    return
    print("kernel not found:", want)


def report(name, lib, body):
    ins = []                      # (address, op, text)
    for l in body:
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", l)
        if m:
            ins.append((int(m.group(3), 16), m.group(1), (m.group(1) + " " + m.group(2)).strip()))
    addr_index = {a: k for k, (a, _, _) in enumerate(ins)}
    # branch targets: s_cbranch / s_branch take a signed 16-bit dword offset relative to the next instruction
    loops = []
    for k, (a, op, text) in enumerate(ins):
        if op.startswith("s_cbranch") or op == "s_branch":
            m = re.search(r"\s(\d+)$", text)
            if not m:
                continue
            off = int(m.group(1)); off = off - 65536 if off >= 32768 else off
            tgt = a + 4 + 4 * off
            if off < 0 and tgt in addr_index:
                loops.append((addr_index[tgt], k))
    def is_iter(a, b):
        ops = [o for _, o, _ in ins[a:b + 1]]
        return any(o.startswith("v_dot4") or o.startswith("v_dot2") or o.startswith("v_mul_i32_i24") or o.startswith("v_mad_i32_i24") for o in ops) and \
            any("permlane32_swap" in o for o in ops) and any(o.startswith("ds_read") for o in ops)
    cand = sorted(set((a, b) for (a, b) in loops if is_iter(a, b)))
    inner = [(a, b) for (a, b) in cand if not any((c, d) != (a, b) and c >= a and d <= b for (c, d) in cand)]
    print("# Instruction mix of `%s` (%s)\n" % (name, os.path.relpath(lib, ROOT)))
    print("Static counts from the gfx950 ISA (`tools/isa_mix.py`).  An LK iteration loop = the smallest backward-branch region that holds the\n"
          "bilinear taps, the LDS reads and the lane-swap reduction; the body is executed once per iteration by every wave of the corner\n"
          "(rare paths -- tile re-staging, the FP64 tie-break -- are inside the region and counted, though seldom executed).\n")

    def table(sel, title):
        cnt = {}
        for _, op, _ in sel:
            c = classify(op); cnt[c] = cnt.get(c, 0) + 1
        tot = len(sel)
        valu = sum(v for c, v in cnt.items() if c in [x[0] for x in CLASSES[:7]])
        print("## %s -- %d instructions, %d VALU\n\n| class | count |\n|---|---|" % (title, tot, valu))
        for c, _ in CLASSES:
            if cnt.get(c):
                print("| %s | %d |" % (c, cnt[c]))
        print()
    table(ins, "whole kernel")
    if PATH:
        sel = [i for i in ins if any(a <= i[0] <= b for a, b in PATH)]
        table(sel, "one trip of the iteration loop, common path (%s)" % ", ".join("0x%x..0x%x" % r for r in PATH))
    for n, (a, b) in enumerate(sorted(inner)):
        table(ins[a:b + 1], "LK iteration loop %d (0x%x..0x%x)" % (n, ins[a][0], ins[b][0]))


if __name__ == "__main__":
    main()
