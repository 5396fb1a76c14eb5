#!/bin/bash
# round 6, GPU session AO: cold pairs with and without the co-tenant issue priority (knobs build, AGT_LK_COTENANT_PRIO), four alternating runs, same box
set -o pipefail
OUT=gpurun_out/r6ao
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
show() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]; print(d["ms_per_step"], d["timing"]["ms_per_step_p10"], r["whole_step"]["frac_of_8TBs"], r["avg_launch_us"], r["alone"]["avg_launch_us"])
except Exception as e: print("parse", e)
PY
}
run() { name=$1; shift; echo "== $name"; timeout -k 10 400 python3 tools/knobbench.py --no-cpu-baseline --workload c3pairs --steps 1024 --warmup 32 "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "rc=$?"; show "$OUT/$name.json"; }
for i in 1 2 3 4; do
AGT_LK_COTENANT_PRIO=1 run prio1_$i
AGT_LK_COTENANT_PRIO=0 run prio0_$i
done
