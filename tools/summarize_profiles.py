#!/usr/bin/env python3
"""Turn a tools/collect_profiles.sh output directory (gpurun_out/<name>) into the files kept under profiles/:
  r02_<run>_kernel_stats.csv   rocprofv3 --stats summary (copied)
  r02_<run>_bench.json         the bench line of the profiled command
  r02_kernel_durations.json    per kernel and run: dispatches, mean / median / p90 duration from the kernel trace; for the
                               fused step kernel only the full launches (grid = the most frequent grid size)
  r02_pmc_raw.json             per kernel and run: mean FETCH_SIZE / WRITE_SIZE (KiB per dispatch)
  pmc_traffic.json             'step_kernel<21,4,3> depth F' entries for the F the profiled commands ran at
Usage: python3 tools/summarize_profiles.py gpurun_out/r3final [r03]      (second argument: file prefix, default r02)"""
import csv, glob, json, os, shutil, sys
from collections import Counter, defaultdict
import numpy as np

src = sys.argv[1]
PFX = sys.argv[2] if len(sys.argv) > 2 else "r02"
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
short = lambda n: n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]

def bench_line(run):
    p = os.path.join(src, run + ".stdout")
    for line in reversed(open(p).read().strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    return None

def full_grid(b, v):
    """grid size of the fused step's FULL launches of a c2-shaped run: (corners + streams + 60 pyramid tiles x F frames) x 256
    threads (the runs also contain one-frame launches of the instrumented passes and the drain launches of every block)"""
    F = (b or {}).get("roofline", {}).get("frames_per_launch")
    B = (b or {}).get("config", {}).get("streams_per_gpu", 1)
    if F and (b or {}).get("config", {}).get("workload", "").startswith("c2"):
        g = (48 * B + B) * 256          # a block of F frames: the pyramid goes out as its own launch, the chained launch carries LK | PnP only
        if sum(1 for x in v if x[1] == g) >= 2 and F == (b or {}).get("steps"):
            return g                    # (checked FIRST: such a run also holds one launch with the pyramid role riding -- rounds 2 / 3 sampled that one)
        g = (48 * B + B + 60 * B * F) * 256
        if any(x[1] == g for x in v):
            return g
    tot = defaultdict(float)
    for x in v: tot[x[1]] += x[0]
    return max(tot, key=tot.get)

durations, pmc = {}, {}
for d in sorted(glob.glob(os.path.join(src, "*_stats"))):
    run = os.path.basename(d)[:-6]
    ks = glob.glob(os.path.join(d, "*", "*_kernel_stats.csv"))[0]
    shutil.copy(ks, os.path.join(dst, PFX + "_%s_kernel_stats.csv" % run))
    b = bench_line(run + "_stats")
    if b:
        json.dump(b, open(os.path.join(dst, PFX + "_%s_bench.json" % run), "w"), indent=1)
    per = defaultdict(list)
    for r in csv.DictReader(open(glob.glob(os.path.join(d, "*", "*_kernel_trace.csv"))[0])):
        per[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Grid_Size_X"])))
    out = {}
    for k, v in per.items():
        if k.startswith("__amd") or len(v) < 5:
            continue
        if k.startswith("step_kernel"):
            g = full_grid(b, v)
            v = [x for x in v if x[1] == g]
        t = np.array([x[0] for x in v], float) / 1e3
        out[k] = {"dispatches": len(t), "mean_us": round(float(t.mean()), 2), "median_us": round(float(np.median(t)), 2),
                  "p90_us": round(float(np.percentile(t, 90)), 2), "grid": Counter(x[1] for x in v).most_common(1)[0][0]}
    durations[run] = {"command_metric": b and {"value": b["value"], "ms_per_step": b["ms_per_step"], "steps": b["steps"],
                                               "launch": b["config"].get("launch")}, "kernels": out}
json.dump(durations, open(os.path.join(dst, PFX + "_kernel_durations.json"), "w"), indent=1)

for kind in ("fetch", "write"):
    for d in sorted(glob.glob(os.path.join(src, "*_" + kind))):
        run = os.path.basename(d)[:-len(kind) - 1]
        per = defaultdict(list)
        for r in csv.DictReader(open(glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))[0])):
            per[short(r["Kernel_Name"])].append((float(r["Counter_Value"]), int(r["Grid_Size"])))
        for k, v in per.items():
            if k.startswith("__amd_rocclr_fill") or len(v) < 2:
                continue
            if k.startswith("step_kernel"):
                g = full_grid(bench_line(run + "_" + kind), v)
                v = [x for x in v if x[1] == g]
            if len(v) < 2:
                continue
            e = pmc.setdefault(run, {}).setdefault(k, {})
            e[("FETCH" if kind == "fetch" else "WRITE") + "_SIZE_KiB_mean"] = round(float(np.mean([x[0] for x in v])), 1)
            e["dispatches_" + kind] = len(v)
        b = bench_line(run + "_" + kind)
        if b and "frames_per_launch" in b.get("roofline", {}):
            pmc[run]["_frames_per_launch"] = b["roofline"]["frames_per_launch"]
json.dump(pmc, open(os.path.join(dst, PFX + "_pmc_raw.json"), "w"), indent=1)

# launch period per depth from the kernel-trace-only runs (the counter passes themselves slow the kernels down by ~30 %)
stats_launch_us = {}
for d in sorted(glob.glob(os.path.join(src, "*_stats"))):
    b = bench_line(os.path.basename(d))
    r = (b or {}).get("roofline", {})
    if (b or {}).get("config", {}).get("workload", "").startswith("c2") and r.get("frames_per_launch"):
        stats_launch_us[r["frames_per_launch"]] = r.get("avg_launch_us")
tp = os.path.join(dst, "pmc_traffic.json")
traffic = json.load(open(tp))
for run, ks in pmc.items():
    F = ks.get("_frames_per_launch")
    for k, e in ks.items():
        if not k.startswith("step_kernel<21, 4, 3") or "FETCH_SIZE_KiB_mean" not in e or "WRITE_SIZE_KiB_mean" not in e or not F:
            continue
        fetch, write, parts, label = e["FETCH_SIZE_KiB_mean"], e["WRITE_SIZE_KiB_mean"], None, "%d frames per chained launch (full launches only)" % F
        b_run = bench_line(run + "_fetch")
        if b_run and b_run.get("steps") == F and "pyr_group_kernel" in ks:
            # driver-style blocks of F frames: per block ONE pyramid launch (pyr_group_kernel) + ONE chained LK | PnP launch; the block's
            # traffic is the sum of the two
            pg = ks["pyr_group_kernel"]
            parts = {"pyr_group_kernel": {"FETCH": pg["FETCH_SIZE_KiB_mean"], "WRITE": pg["WRITE_SIZE_KiB_mean"]},
                     "step_kernel (LK | PnP)": {"FETCH": fetch, "WRITE": write}}
            fetch += pg["FETCH_SIZE_KiB_mean"]; write += pg["WRITE_SIZE_KiB_mean"]
            label = "blocks of %d frames (driver style): per block ONE pyramid launch (pyr_group_kernel) + ONE chained LK | PnP launch (step_kernel, 49 workgroups); the figures are the sum of the two" % F
        traffic["step_kernel<21,4,3> depth %d" % F] = {
            "workload": "c2: 1 x 1280x720 stream, " + label, "parts_KiB": parts,
            "dispatches": e["dispatches_fetch"], "FETCH_SIZE_KiB_mean": round(fetch, 1), "WRITE_SIZE_KiB_mean": round(write, 1),
            "traffic_bytes_per_launch": int(round((2 * fetch + write) * 1024)),
            "algorithmic_bytes_per_launch": 1441008 * F, "build": PFX + " final (chained launch, frame-chained LK role)",
            # launch period bench.py measured (HIP events) in the run the counters were taken in: bench.py withholds the figure
            # when its own launch period has moved away from this by more than 15 %
            "launch_us_at_collection": stats_launch_us.get(F)}
def raw_entry(run, prefix, name, label, alg=None):
    """pmc_traffic.json entry `name` from the raw counters of the first kernel of `run` whose short name starts with `prefix`"""
    for k, e in pmc.get(run, {}).items():
        if k.startswith(prefix) and "FETCH_SIZE_KiB_mean" in e and "WRITE_SIZE_KiB_mean" in e:
            us = durations.get(run, {}).get("kernels", {}).get(k, {}).get("mean_us")
            traffic[name] = {"workload": label, "kernel": k, "dispatches": e["dispatches_fetch"], "FETCH_SIZE_KiB_mean": e["FETCH_SIZE_KiB_mean"],
                             "WRITE_SIZE_KiB_mean": e["WRITE_SIZE_KiB_mean"],
                             "traffic_bytes_per_launch": int(round((2 * e["FETCH_SIZE_KiB_mean"] + e["WRITE_SIZE_KiB_mean"]) * 1024)),
                             "algorithmic_bytes_per_launch": alg, "build": PFX, "kernel_trace_mean_us": us, "launch_us_at_collection": None,
                             "note": "mean over every dispatch of the kernel in the run; FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM section)"}
            return
raw_entry("c5", "dense_accum_kernel", "dense_accum_kernel", "c5: 61,440 samples + 240 corners, one Gauss-Newton launch (update prologue + accumulate)", 61440 * 28)
raw_entry("c3pairs", "pyr_roll2_kernel", "pyr_roll2_kernel c3pairs", "c3pairs: two pyrDown levels per pass (register-rolling, alternating strip directions) over 64 cold 720p frames, "
          "one launch per pyramid build; algorithmic bytes = 64 x W*H*1.3125 (SURVEY 8d: level 0 read once, levels 1 and 2 written once)", int(64 * 1280 * 720 * 1.3125))
raw_entry("c3pairs", "pyr_group_kernel", "pyr_group_kernel c3pairs (pair build)", "c3pairs, round 6 (agt_pyramid_build_pair): the two-level rolling pass over BOTH frames of 64 cold 720p pairs = 128 frames "
          "in one launch; algorithmic bytes = 128 x W*H*1.3125 (SURVEY 8d)", int(128 * 1280 * 720 * 1.3125))
raw_entry("c3pairs", "lk_kernel<21, 1, 3", "lk_kernel<21,1,3> c3pairs", "c3pairs: 3072 corners, one wave per corner", 64 * (48 * 3 * 1600 + 48 * 21))
json.dump(traffic, open(tp, "w"), indent=1)
print(json.dumps({r: {k: v for k, v in d["kernels"].items()} for r, d in durations.items()}, indent=1)[:6000])
print(json.dumps({k: v for k, v in traffic.items() if "depth" in k}, indent=1))
