#!/usr/bin/env python3
"""ISA-level account of ONE Levenberg-Marquardt evaluation of the pose solver: the instructions between the in-kernel
stamps 48 .. 51 of the diagnostic library's pnp_kernel<float, 1> (agt_pnp_body.h evaluate_t, mode 2: Rodrigues with derivative |
per-point projection + 2 x 6 Jacobian + the 28 products | 28-sum register butterfly), classified, with the length of the longest
register-dependency chain of every part.

    python3 tools/pnp_eval_isa.py [libagt_hip_dbg.so] > profiles/r05_pnp_evaluation_isa.md

The stamps are `s_memtime` reads whose results thread 0 stores at agt_pnp_stamps[i] (offset 8 i); the region of stamp i .. i + 1 is the
straight-line code between the two reads.  Times per part come from tools/pnpstamps.py on the GPU (one wave alone on its SIMD)."""
import os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa_mix import code_objects, LLVM, ROOT

CLASSES = [
    ("FP64 fma / mul / add", r"^v_(fma|mul|add|fmac)_f64"),
    ("FP64 rcp / rsq / sqrt / div helpers / trig", r"^v_(rcp|rsq|sqrt|div_scale|div_fmas|div_fixup|fract|trig_preop|ldexp|frexp\w*|rndne|floor|ceil|trunc)_f64"),
    ("FP64 compare / select / min / max / cvt", r"^v_(cmp\w*_f64|cmpx\w*_f64|min_f64|max_f64|cvt_\w*f64\w*|cvt_f64\w*|cndmask_b32)"),
    ("DPP / permlane / swaps (butterfly moves)", r"(_dpp$|^v_permlane|^v_mov_b32_dpp|^v_mov_b64_dpp|^v_swap)"),
    ("readlane / readfirstlane / writelane", r"^v_(read(first)?lane|writelane)"),
    ("moves (v_mov, accvgpr)", r"^v_(mov_b32|mov_b64|accvgpr_(read|write|mov))"),
    ("FP32 / integer / logic VALU", r"^v_"),
    ("LDS", r"^ds_"), ("global / buffer / flat / scratch memory", r"^(global_|buffer_|flat_|scratch_)"),
    ("scalar ALU / moves", r"^s_(?!waitcnt|barrier|cbranch|branch|nop|endpgm|load|buffer_load|sleep|setprio|memtime|memrealtime)"),
    ("scalar memory", r"^s_(load|buffer_load|memtime|memrealtime)"), ("waitcnt", r"^s_waitcnt"), ("branch", r"^s_(c)?branch"),
    ("barrier / nop / other", r"."),
]


def classify(op):
    for name, pat in CLASSES:
        if re.search(pat, op):
            return name
    return "barrier / nop / other"


def regs(tok):
    """register names an operand token touches: v12 -> [v12]; v[4:5] -> [v4, v5]; s[2:3]; a12; vcc, exec, scc"""
    tok = tok.strip().rstrip(",")
    tok = re.sub(r"^(-|\|)+|\|+$", "", tok)
    tok = re.sub(r"^(neg|abs|sext)\((.*)\)$", r"\2", tok)
    m = re.match(r"^([vsa])\[(\d+):(\d+)\]$", tok)
    if m:
        return ["%s%d" % (m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)]
    if re.match(r"^[vsa]\d+$", tok):
        return [tok]
    if tok in ("vcc", "vcc_lo", "vcc_hi"):
        return ["vcc"]
    if tok in ("exec", "exec_lo", "exec_hi"):
        return ["exec"]
    if tok == "scc":
        return ["scc"]
    return []


def parse(line):
    line = line.split("//")[0].strip()
    if not line or line.endswith(":"):
        return None
    parts = line.split(None, 1)
    op = parts[0]
    ops = []
    if len(parts) > 1:
        depth, cur = 0, ""
        for ch in parts[1]:
            if ch == "[":
                depth += 1
            if ch == "]":
                depth -= 1
            if ch == "," and depth == 0:
                ops.append(cur); cur = ""
            else:
                cur += ch
        ops.append(cur)
    ops = [o.strip().split()[0] if o.strip() else "" for o in ops]        # drop modifiers (row_shr:1, op_sel...)
    return op, ops


def dests_sources(op, ops):
    """(written registers, read registers) of one instruction -- first operand written (two for the 64-bit-carry forms), rest read;
    stores, compares into vcc and scalar compares are handled by name"""
    if not ops:
        return [], []
    if re.match(r"^(global_store|buffer_store|flat_store|scratch_store|ds_write|ds_add|s_waitcnt|s_nop|s_barrier|s_cbranch|s_branch)", op):
        return [], sum((regs(o) for o in ops), [])
    if op.startswith("v_cmp") or op.startswith("v_cmpx"):
        d = regs(ops[0]) or ["vcc"]
        return d, sum((regs(o) for o in ops[1:]), [])
    if op.startswith("s_cmp") or op.startswith("s_bitcmp"):
        return ["scc"], sum((regs(o) for o in ops), [])
    d = regs(ops[0])
    srcs = sum((regs(o) for o in ops[1:]), [])
    if op.startswith("v_cndmask") and len(ops) < 4:
        srcs.append("vcc")
    if op in ("v_div_scale_f64", "v_mad_u64_u32", "v_mad_i64_i32", "v_add_co_u32", "v_sub_co_u32", "v_addc_co_u32"):
        d += regs(ops[1]); srcs = sum((regs(o) for o in ops[2:]), [])
    if op.endswith("_dpp") or op.startswith("v_fmac") or op.startswith("v_mac") or op.startswith("v_writelane"):
        srcs += d                                # (the old destination value is an input)
    if op.startswith("s_cselect") or op.startswith("s_cbranch_scc"):
        srcs.append("scc")
    return d, srcs


def hot_path(lines):
    """the lines of a region without its COLD block: a forward conditional branch whose skipped span holds v_trig_preop_f64 -- the
    inline copy of the library's general-argument sincos (Payne-Hanek reduction), entered for rotation angles >= 2^20 only
    (agt_device.h agt_sincos) -- is assumed taken"""
    addr = []
    for ln in lines:
        m = re.search(r"//\s*([0-9A-Fa-f]{6,16}):", ln)
        addr.append(int(m.group(1), 16) if m else None)
    out, i = [], 0
    while i < len(lines):
        ln = lines[i]
        out.append(ln)
        m = re.match(r"\s*s_cbranch_\w+\s+\d+\s*//.*<.*\+0x([0-9a-fA-F]+)>", ln)
        if m and addr[i] is not None:
            tgt_off = int(m.group(1), 16)
            # target line: first line at or past the target offset (offsets are relative to the kernel's first instruction)
            base = hot_path.base
            j = i + 1
            while j < len(lines) and (addr[j] is None or addr[j] - base < tgt_off):
                j += 1
            if j <= len(lines) and any("v_trig_preop_f64" in x for x in lines[i + 1:j]):
                i = j
                continue
        i += 1
    return out


def region_stats(lines):
    counts, chain, valu, n = {}, {}, 0, 0
    longest = 0
    for ln in lines:
        p = parse(ln)
        if not p:
            continue
        op, ops = p
        n += 1
        c = classify(op)
        counts[c] = counts.get(c, 0) + 1
        if op.startswith("v_"):
            valu += 1
        d, s = dests_sources(op, ops)
        depth = 1 + max([chain.get(r, 0) for r in s] + [0])
        if op.startswith("s_waitcnt") or op.startswith("s_nop"):
            continue
        for r in d:
            chain[r] = depth
        longest = max(longest, depth)
    return n, valu, counts, longest


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "accurate_aprilgroup_tracking_amd", "libagt_hip_dbg.so")
    want = "pnp_kernel<float, 1>"
    body = None
    with tempfile.TemporaryDirectory() as tmp:
        for co in code_objects(lib, tmp):
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
            dem = subprocess.run(["c++filt"], input=dis, capture_output=True, text=True).stdout.splitlines()
            starts = [i for i, l in enumerate(dem) if re.match(r"^[0-9a-f]+ <.*>:$", l)]
            for si, i in enumerate(starts):
                if want in dem[i]:
                    body = dem[i + 1:(starts[si + 1] if si + 1 < len(starts) else len(dem))]
    if body is None:
        print("kernel not found", file=sys.stderr); sys.exit(1)
    # stamp i is marked by `s_mov_b32 sN, 0xbeef00 + i` (agt_pnp_body.h PSTAMPM); the evaluation is inlined at several call sites --
    # every complete 48 .. 51 run is reported
    stamp_at = {}
    for i, l in enumerate(body):
        m = re.search(r"s_mov_b32\s+s\d+,\s*0xbeef([0-9a-f]{2})\b", l)
        if m:
            stamp_at.setdefault(int(m.group(1), 16), []).append(i)
    runs = []
    for a in stamp_at.get(48, []):
        try:
            b = min(t for t in stamp_at.get(49, []) if t > a)
            c = min(t for t in stamp_at.get(50, []) if t > b)
            d = min(t for t in stamp_at.get(51, []) if t > c)
        except ValueError:
            continue
        runs.append((a, b, c, d))
    m0 = None
    for ln in body:
        m0 = re.search(r"//\s*([0-9A-Fa-f]{6,16}):", ln)
        if m0:
            break
    hot_path.base = int(m0.group(1), 16) if m0 else 0
    print("# One Levenberg-Marquardt evaluation of the pose solver, instruction by instruction (`pnp_kernel<float, 1>`, diagnostic library)\n")
    print("`tools/pnp_eval_isa.py`: the code between the in-kernel stamps 48 | 49 | 50 | 51 of `agt_pnp_body.h evaluate_t` (mode 2: the")
    print("evaluation that also leaves J^T J / J^T e for the next iteration).  One point per lane, 48 of 64 lanes active, one wave per problem.")
    print("`chain` = the longest register-dependency chain of the part, in instructions (waitcnt / nop excluded).\n")
    names = ["Rodrigues (R and dR/dr)", "projection + 2 x 6 Jacobian + 28 products (per point = per lane)", "28-sum register butterfly (permlane / DPP / adds)"]
    for k, (a, b, c, d) in enumerate(runs):
        print("## inlined instance %d (ISA lines %d .. %d)\n" % (k, a, d))
        print("| part | instructions | VALU | longest chain | " + " | ".join(n for n, _ in CLASSES) + " |")
        print("|---|---|---|---|" + "---|" * len(CLASSES))
        tot = [0, 0, 0]
        for nm, (lo, hi) in zip(names, ((a, b), (b, c), (c, d))):
            reg = body[lo + 1:hi]
            hot = hot_path(reg)
            n, valu, counts, longest = region_stats(hot)
            tot[0] += n; tot[1] += valu; tot[2] += longest
            cold = len([1 for x in reg if parse(x)]) - n
            print("| %s%s | %d | %d | %d | " % (nm, (" -- %d more instructions in cold blocks, not counted" % cold) if cold else "", n, valu, longest) +
                  " | ".join(str(counts.get(cn, 0)) for cn, _ in CLASSES) + " |")
        print("| **whole evaluation** | %d | %d | %d (sum of the parts: they are sequential) | |\n" % tuple(tot))


if __name__ == "__main__":
    main()
