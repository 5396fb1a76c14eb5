#!/bin/bash
# round 5, GPU session AJ: cold-pair step with the pose solves on one extra highest-priority stream
set -o pipefail
OUT=gpurun_out/r5aj
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
show() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["timing"]["ms_per_step_p10"], d["max_abs_pose_err_vs_truth"])
except Exception as e: print("parse", e)
PY
}
run() { name=$1; shift; echo "== $name"; timeout -k 10 300 python3 bench.py --workload c3pairs --steps 1024 --no-cpu-baseline "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "rc=$?"; show "$OUT/$name.json"; }
for i in 1 2; do
run same4_$i
run high4_$i --pnp-stream high
run high3_$i --pnp-stream high --pair-contexts 3
run high5_$i --pnp-stream high --pair-contexts 5
done
