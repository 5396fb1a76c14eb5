#!/bin/bash
# round 5, GPU session AQ: pyramid-only launch at the head of a run in the rolling two-level form: whole suite, c2 20-step blocks against the tree before
set -o pipefail
OUT=gpurun_out/r5aq
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
timeout -k 10 1000 python3 -m pytest tests -q -m gpu -x > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest.log"
show() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["timing"]["ms_per_step_p10"])
except Exception as e: print("parse", e)
PY
}
run() { name=$1; dir=$2; shift 2; echo "== $name"; (cd $dir && timeout -k 10 400 "$@") > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "rc=$?"; show "$OUT/$name.json"; }
A="--steps 20 --warmup 5 --blocks 45 --no-cpu-baseline --no-extras"
for i in 1 2 3; do
run c2k20_new_$i . python3 bench.py $A
run c2k20_prev_$i prevtree python3 bench.py $A
done
run c2_new . python3 bench.py --no-cpu-baseline --no-extras
run c2_prev prevtree python3 bench.py --no-cpu-baseline --no-extras
run c4k20_new . python3 bench.py --workload c4 --steps 20 --warmup 5 --blocks 45 --no-cpu-baseline --no-extras
run c4k20_prev prevtree python3 bench.py --workload c4 --steps 20 --warmup 5 --blocks 45 --no-cpu-baseline --no-extras
