#!/usr/bin/env python3
"""Counters of the tools/fetchcal passes beside the byte counts the program printed:
python3 tools/fetchcal_summary.py gpurun_out/<dir> > profiles/r05_fetch_size_calibration.json"""
import collections, csv, glob, json, os, sys
d = sys.argv[1]
exp = json.load(open(os.path.join(d, "fetchcal_expected.json")))
out = {"_what": "rocprofv3 --pmc <counter> --kernel-trace -- ./tools/fetchcal 3 (one counter per pass, MI355X); per kernel: the counter's mean "
                "over 3 dispatches, and the counter as bytes (FETCH_SIZE x 1024; request counters x 64) divided by the bytes the pattern "
                "touches -- exactly, in whole 32-B / 64-B sectors, in whole 128-B lines",
       "expected_bytes": {k: v for k, v in exp.items() if isinstance(v, dict)}, "counters": {}}
for path in sorted(glob.glob(os.path.join(d, "cal_*/*/*_counter_collection.csv"))):
    acc = collections.defaultdict(list)
    cname = None
    for r in csv.DictReader(open(path)):
        cname = r["Counter_Name"]
        acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    scale = 1024.0 if cname == "FETCH_SIZE" else 64.0
    ent = {}
    for k, v in acc.items():
        if k not in exp:
            continue
        m = sum(v) / len(v)
        ent[k] = {"mean": round(m, 1), "dispatches": len(v)}
        if cname in ("FETCH_SIZE", "TCC_EA0_RDREQ_sum", "TCC_MISS_sum", "TCC_REQ_sum"):
            ent[k]["as_bytes_over"] = {kk: round(m * scale / vv, 4) for kk, vv in exp[k].items()}
    out["counters"][cname] = ent
out["conclusion"] = ("FETCH_SIZE = TCC_EA0_RDREQ x 64 B, and the L2 issues ONE read request per 128-BYTE LINE it misses, whatever the width of the "
                     "load that missed (b128 / b64 / b32 streams all read 0.500 of their bytes; dword tile rows, 44 and 28 bytes long, and whole "
                     "64-B sectors fetched with b128 loads all read 0.503 of their 128-B-line footprint; TCC_EA0_RDREQ_32B is 0).  So 2 x "
                     "FETCH_SIZE is the memory-side read traffic in whole 128-B lines for EVERY pattern of this code base, the dword tile loads "
                     "of the LK kernels included: the LK figure carried as 'uncalibrated upper bound' in rounds 1-4 (48.9 MB per 3,072-corner "
                     "launch against 14.8 MB algorithmic) is the real line traffic.  A 44-byte tile row costs one or two whole lines.")
json.dump(out, sys.stdout, indent=1)
