#!/usr/bin/env python3
"""Where the time of a pipelined c3 block goes (rocprofv3 --kernel-trace CSV of `bench.py --workload c3`): blocks are separated by the
host's synchronisation (gaps > 300 us on the LK queues); inside a block: LK queue busy time, the gaps on the LK queues with the kernels
of the other queues that ran during them, kernel durations.  python tools/trace_blocks.py <kernel_trace.csv>"""
import csv, sys, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
name = lambda r: r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], name(r)) for r in rows)
lkq = sorted({q for s, e, q, n in ks if n.startswith("lk_kernel")}, key=lambda q: -sum(1 for k in ks if k[2] == q and k[3].startswith("lk_kernel")))[:2]
lk = [k for k in ks if k[2] in lkq and k[3].startswith("lk_kernel")]
# split into blocks at long silences of both LK queues
blocks, cur = [], [lk[0]]
for k in lk[1:]:
    if k[0] - max(c[1] for c in cur[-4:]) > 300e3: blocks.append(cur); cur = []
    cur.append(k)
blocks.append(cur)
print("%d LK launches on queues %s, %d blocks" % (len(lk), lkq, len(blocks)))
for bi, b in enumerate(blocks):
    if len(b) < 64: continue
    t0, t1 = b[0][0], max(k[1] for k in b)
    others = [k for k in ks if k[2] not in lkq and k[1] > t0 - 2e6 and k[0] < t1 + 2e6 and not k[3].startswith("__amd")]
    first = min([k[0] for k in others if k[0] > t0 - 1.5e6] + [t0]); last = max([k[1] for k in others if k[1] < t1 + 1.5e6] + [t1])
    frames = len(b) / 2
    print("block %d: %d frames | first kernel of the block .. last: %.0f us = %.2f us per frame | LK window %.0f us = %.2f per frame | before LK %.0f us, after LK %.0f us"
          % (bi, frames, (last - first) / 1e3, (last - first) / 1e3 / frames, (t1 - t0) / 1e3, (t1 - t0) / 1e3 / frames, (t0 - first) / 1e3, (last - t1) / 1e3))
    for q in lkq:
        kq = [k for k in b if k[2] == q]
        d = [(k[1] - k[0]) / 1e3 for k in kq]; g = [(kq[i + 1][0] - kq[i][1]) / 1e3 for i in range(len(kq) - 1)]
        print("   queue %s: %d launches, duration mean %.1f p50 %.1f p90 %.1f | gaps: total %.0f us, %d over 3 us (sum %.0f)" % (
            q, len(kq), statistics.mean(d), statistics.median(d), sorted(d)[len(d) * 9 // 10], sum(g), sum(1 for x in g if x > 3), sum(x for x in g if x > 3)))
    for n in sorted({k[3] for k in others}):
        d = [(k[1] - k[0]) / 1e3 for k in others if k[3] == n and k[0] >= first and k[1] <= last]
        if d: print("   %-28s %3d launches, mean %.1f us, total %.0f us" % (n[:28], len(d), statistics.mean(d), sum(d)))
