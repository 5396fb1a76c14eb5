#!/bin/bash
# round 6, GPU session L: agt_pyramid_build_pair -- parity tests, then the cold-pair step with one pair build against two builds, same box
set -o pipefail
OUT=gpurun_out/r6l
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1; rc=$?; tail -5 "$OUT/pytest.log"; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit $rc
show() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]; print(d["value"], d["ms_per_step"], d["timing"]["ms_per_step_p10"], r["whole_step"]["frac_of_8TBs"], r["frac"], r["avg_launch_us"], r["alone"]["avg_launch_us"], r["alone"]["frac"])
except Exception as e: print("parse", e)
PY
}
run() { name=$1; shift; echo "== $name"; timeout -k 10 400 python3 bench.py --no-cpu-baseline --workload c3pairs --steps 256 "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "rc=$?"; show "$OUT/$name.json"; }
for i in 1 2 3; do
run pair_$i
run two_$i --no-pair-build
done
