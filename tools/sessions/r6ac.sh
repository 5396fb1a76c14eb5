#!/bin/bash
# round 6, GPU session AC: after the LDS request rule changed (ABI 505: smallest multiple of 256 B above LDS / (n + 1)): cap test, c3 and cold pairs
set -o pipefail
OUT=gpurun_out/r6ac
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "occupancy" > "$OUT/pytest.log" 2>&1; rc=$?; tail -3 "$OUT/pytest.log"; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit $rc
show() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]; print(d["ms_per_step"], d["timing"]["ms_per_step_p10"], r["whole_step"]["frac_of_8TBs"], r["avg_launch_us"], d.get("accepted_frac"))
except Exception as e: print("parse", e)
PY
}
run() { name=$1; shift; echo "== $name"; timeout -k 10 300 python3 bench.py --no-cpu-baseline "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "rc=$?"; show "$OUT/$name.json"; }
for i in 1 2; do
run c3_$i --workload c3 --steps 600
run c3_none_$i --workload c3 --steps 600 --stream-lk-cu 0
run pairs_$i --workload c3pairs --steps 1024 --warmup 32
run pairs_cu9_$i --workload c3pairs --steps 1024 --warmup 32 --lk-cu 9
done
