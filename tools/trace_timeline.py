#!/usr/bin/env python3
"""Timeline of a rocprofv3 --kernel-trace CSV: per queue, the chain of kernels with start / end / gap to the previous kernel of the
queue, for a window of the steady state; and per kernel name duration statistics.  python tools/trace_timeline.py <kernel_trace.csv> [t0_us] [span_us]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
name = lambda r: r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), name(r)) for r in rows))
tmin = ks[0][0]
T0 = float(sys.argv[2]) if len(sys.argv) > 2 else None
SPAN = float(sys.argv[3]) if len(sys.argv) > 3 else 200.0
by = collections.defaultdict(list)
for s, e, q, n in ks: by[n].append((e - s) / 1e3)
print("kernel durations (us): count mean p10 p50 p90")
import statistics
for n, d in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:10]:
    d = sorted(d); print("  %-40s %6d %8.2f %8.2f %8.2f %8.2f" % (n[:40], len(d), statistics.mean(d), d[len(d) // 10], d[len(d) // 2], d[len(d) * 9 // 10]))
# steady-state window: the middle of the trace
if T0 is None: T0 = ((ks[len(ks) // 2][0] - tmin) / 1e3)
last = {}
print("window %.1f .. %.1f us: start end dur gap-on-queue queue kernel" % (T0, T0 + SPAN))
for s, e, q, n in ks:
    su, eu = (s - tmin) / 1e3, (e - tmin) / 1e3
    if T0 <= su <= T0 + SPAN:
        g = su - last[q] if q in last else float("nan")
        print("  %9.1f %9.1f %7.1f %7.1f  q%-3s %s" % (su - T0, eu - T0, eu - su, g, q, n[:50]))
    last[q] = eu
# gaps per (queue, kernel) between consecutive kernels of the same queue
gaps = collections.defaultdict(list); last = {}
for s, e, q, n in ks:
    if q in last: gaps[(q, n)].append((s - last[q]) / 1e3)
    last[q] = e
print("gap before a kernel on its queue (us): queue kernel count p50 p90")
for (q, n), g in sorted(gaps.items(), key=lambda kv: -len(kv[1]))[:12]:
    g = sorted(g); print("  q%-3s %-40s %6d %8.2f %8.2f" % (q, n[:40], len(g), g[len(g) // 2], g[len(g) * 9 // 10]))
