#!/usr/bin/env python3
"""Engine / memory clock and power of GPU 0 while a bench command runs (rocm-smi polled every 0.25 s in this process, the bench as a child):
does the chip hold its clock under the 64-stream step?  python3 tools/clockwatch.py <seconds> -- python3 bench.py --workload c3 ...
Prints one JSON object: samples while the child ran (min / median / max of sclk, mclk, power) and the idle reading before it."""
import json, re, subprocess, sys, time

def sample():
    try:
        out = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showpower", "--showuse"], capture_output=True, text=True, timeout=10).stdout
    except Exception as e:
        return {"error": repr(e)}
    r = {}
    m = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", out); r["sclk"] = int(m.group(1)) if m else None
    m = re.search(r"mclk clock level: \S+ \((\d+)Mhz\)", out); r["mclk"] = int(m.group(1)) if m else None
    m = re.search(r"Power \(W\): ([\d.]+)", out) or re.search(r"Socket Power \(W\): ([\d.]+)", out); r["power"] = float(m.group(1)) if m else None
    m = re.search(r"GPU use \(%\): (\d+)", out); r["use"] = int(m.group(1)) if m else None
    return r

def stats(xs):
    xs = sorted(x for x in xs if x is not None)
    return None if not xs else {"min": xs[0], "median": xs[len(xs) // 2], "max": xs[-1], "n": len(xs)}

def main():
    i = sys.argv.index("--")
    cmd = sys.argv[i + 1:]
    idle = sample()
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    rows = []
    while p.poll() is None:
        rows.append(sample()); time.sleep(0.25)
    out = p.stdout.read()
    line = [l for l in out.splitlines() if l.startswith("{")]
    res = {}
    if line:
        d = json.loads(line[-1]); res = {"value": d.get("value"), "ms_per_step": d.get("ms_per_step")}
    busy = [r for r in rows if (r.get("use") or 0) >= 50]
    print(json.dumps({"cmd": " ".join(cmd), "idle": idle, "all": {k: stats([r.get(k) for r in rows]) for k in ("sclk", "mclk", "power", "use")},
                      "while_gpu_use_ge_50": {k: stats([r.get(k) for r in busy]) for k in ("sclk", "mclk", "power")}, "bench": res, "raw_first": rows[:3]}))

if __name__ == "__main__":
    main()
