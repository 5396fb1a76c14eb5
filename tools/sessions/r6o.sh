#!/bin/bash
# round 6, GPU session O: the LK role of the split pipeline as ONE launch per group (in-kernel frame loop, one wave per corner: knobs build,
# AGT_SPLIT_LK_GROUP=2; 64 B of scratch today against 240 B in round 4) against the per-frame half-batch launches, c3, same box
set -o pipefail
OUT=gpurun_out/r6o
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
show() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]; print(d["value"], d["ms_per_step"], d["timing"]["ms_per_step_p10"], r["whole_step"]["frac_of_8TBs"], r["frac"], r["avg_launch_us"], d.get("accepted_frac"))
except Exception as e: print("parse", e)
PY
}
run() { name=$1; shift; echo "== $name"; timeout -k 10 300 python3 tools/knobbench.py --no-cpu-baseline --workload c3 --steps 256 "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "rc=$?"; show "$OUT/$name.json"; }
for i in 1 2; do
AGT_SPLIT_LK_GROUP=1 run frame_$i
AGT_SPLIT_LK_GROUP=2 run group_$i
done
