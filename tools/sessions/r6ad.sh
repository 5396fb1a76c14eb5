#!/bin/bash
# round 6, GPU session AD: cold pairs with the pose solves of all batches on ONE more context / stream (events between), 3 + 1 and 4 + 1 streams,
# against the shipped form (4 contexts, solve in the batch's own stream), same box
set -o pipefail
OUT=gpurun_out/r6ad
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
show() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]; print(d["ms_per_step"], d["timing"]["ms_per_step_p10"], r["whole_step"]["frac_of_8TBs"], d.get("max_abs_pose_err_vs_truth"), d.get("tracked_frac"))
except Exception as e: print("parse", e)
PY
}
run() { name=$1; shift; echo "== $name"; timeout -k 10 300 python3 bench.py --no-cpu-baseline --workload c3pairs --steps 1024 --warmup 32 "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "rc=$?"; show "$OUT/$name.json"; }
for i in 1 2; do
run base_$i
run side3_$i --pair-contexts 3 --pnp-stream
run side4_$i --pair-contexts 4 --pnp-stream
run side2_$i --pair-contexts 2 --pnp-stream
done
