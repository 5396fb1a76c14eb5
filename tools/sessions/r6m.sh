#!/bin/bash
# round 6, GPU session M: strip height of the pair build (128 images in one launch: the plan picks 16 level-2 rows; 64 images: 10)
set -o pipefail
OUT=gpurun_out/r6m
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
show() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]; print(d["value"], d["ms_per_step"], d["timing"]["ms_per_step_p10"], r["whole_step"]["frac_of_8TBs"], r["frac"], r["avg_launch_us"], r["alone"]["avg_launch_us"], r["alone"]["frac"])
except Exception as e: print("parse", e)
PY
}
run() { name=$1; shift; echo "== $name"; timeout -k 10 400 python3 tools/knobbench.py --no-cpu-baseline --workload c3pairs --steps 256 "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "rc=$?"; show "$OUT/$name.json"; }
for i in 1 2; do
for oh in 0 6 8 10 12 16; do
AGT_PYR4_OH=$oh run pair_oh${oh}_$i
done
AGT_PYR4_OH=0 run two_$i --no-pair-build
done
