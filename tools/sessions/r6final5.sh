#!/bin/bash
# round 6, last GPU session: the whole GPU suite once, smoke(), the default bench line (as the driver runs it)
set -o pipefail
OUT=gpurun_out/r6final5
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1; rc=$?; tail -4 "$OUT/pytest.log"; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; rc=$?; tail -2 "$OUT/smoke.log"; echo "smoke rc=$rc"
[ $rc -ne 0 ] && exit $rc
t0=$(date +%s)
timeout -k 10 900 python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc=$? in $(( $(date +%s) - t0 )) s"
python3 - "$OUT/bench_default.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("headline", d["metric"], d["value"], d["unit"], "steps", d["steps"], "ms/step", d["ms_per_step"], "roofline", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], "cpu", d["cpu_baseline"]["value"] if d.get("cpu_baseline") else None)
b=d.get("batch64_hbm",{}); print("batch64", b.get("ms_per_step"), b.get("whole_step_frac_of_8TBs"), b.get("blocks_of_256"))
p=d.get("pairs64_hbm",{}); r=p.get("roofline",{}); print("pairs64", p.get("ms_per_step"), r.get("whole_step",{}).get("frac_of_8TBs"), r.get("frac"), (r.get("alone") or {}).get("frac"), p.get("error"))
print("c5", d.get("c5_dense240",{}).get("value"))
PY
