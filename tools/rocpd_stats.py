#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 run kept as a rocpd SQLite database (the default output of this ROCm's rocprofv3
--kernel-trace): calls, total / mean / min / max duration, grid, registers.

    python tools/rocpd_stats.py gpurun_out/.../p_results.db [--grid] > profiles/...
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    by_grid = "--grid" in sys.argv
    key = "name, grid_x, workgroup_x" if by_grid else "name"
    rows = db.execute("select %s, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), "
                      "max(lds_size), max(scratch_size) from kernels group by %s order by sum(duration) desc" % (key, key)).fetchall()
    tot = sum(r[-9] for r in rows)
    span = db.execute("select min(start), max(end) from kernels").fetchone()
    print("# %s: %d dispatches, kernel time %.3f ms, first start to last end %.3f ms" % (sys.argv[1], sum(r[-10] for r in rows), tot / 1e6, (span[1] - span[0]) / 1e6))
    print("# calls  total_ms    %%   avg_us   min_us   max_us  vgpr agpr sgpr    lds scratch  %skernel" % ("grid wg " if by_grid else ""))
    for r in rows:
        name = r[0]
        g = ("%7d %4d " % (r[1], r[2])) if by_grid else ""
        c, s, a, mn, mx, vg, ag, sg, lds, scr = r[-10:]
        print("%6d %9.3f %5.1f %8.2f %8.2f %8.2f  %4d %4d %4d %6d %7d  %s%s" % (c, s / 1e6, 100.0 * s / tot, a / 1e3, mn / 1e3, mx / 1e3, vg, ag, sg, lds, scr, g, name[:140]))


if __name__ == "__main__":
    main()
