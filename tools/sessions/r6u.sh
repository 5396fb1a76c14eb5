#!/bin/bash
# round 6, GPU session U: block length of the 64-stream measurement (a block ends with a join: ~20 frames of pose solves drain alone,
# 4 per launch; 256-step blocks carry that once per 256 frames): 256 / 1024 / 2048 steps per block (no detector refresh inside a block of <= 1,200 steps: corners drift, accepted_frac < 1), same box
set -o pipefail
OUT=gpurun_out/r6u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
show() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]; print(d["ms_per_step"], d["timing"]["ms_per_step_p10"], r["whole_step"]["frac_of_8TBs"], r["avg_launch_us"], d.get("accepted_frac"), d.get("mean_lm_iters"))
except Exception as e: print("parse", e)
PY
}
run() { name=$1; shift; echo "== $name"; timeout -k 10 400 python3 bench.py --no-cpu-baseline "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "rc=$?"; show "$OUT/$name.json"; }
for i in 1 2; do
run c3_256_$i --workload c3 --steps 256
run c3_1024_$i --workload c3 --steps 1024
run c3_2048_$i --workload c3 --steps 2048
done
