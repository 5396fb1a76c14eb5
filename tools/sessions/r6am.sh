#!/bin/bash
# round 6, GPU session AH: strip height of the pyramid role of the split pipeline (16-frame group launches: 1,024 images, the plan caps at 16) 16 / 14 / 12 / 10, c3
set -o pipefail
OUT=gpurun_out/r6am
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
show() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]; print(d["ms_per_step"], d["timing"]["ms_per_step_p10"], r["whole_step"]["frac_of_8TBs"], r["avg_launch_us"], d.get("accepted_frac"))
except Exception as e: print("parse", e)
PY
}
run() { name=$1; shift; echo "== $name"; timeout -k 10 400 python3 tools/knobbench.py --no-cpu-baseline --workload c3 --steps 600 "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "rc=$?"; show "$OUT/$name.json"; }
for i in 1 2 3; do
for oh in 4 6 8; do
AGT_PYR4_OH=$oh run pair_oh${oh}_$i
done
done
