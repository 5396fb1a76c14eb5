#!/bin/bash
# round 6, GPU session AQ: queue priority of the library's LK / pose streams (highest since round 2) against default / lowest, c3, knobs build, same box
set -o pipefail
OUT=gpurun_out/r6aq
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
show() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]; print(d["ms_per_step"], d["timing"]["ms_per_step_p10"], r["whole_step"]["frac_of_8TBs"], d.get("accepted_frac"))
except Exception as e: print("parse", e)
PY
}
run() { name=$1; shift; echo "== $name"; timeout -k 10 300 python3 tools/knobbench.py --no-cpu-baseline --workload c3 --steps 600 "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "rc=$?"; show "$OUT/$name.json"; }
for i in 1 2 3; do
run high_$i
AGT_MS_PRIO=mid run mid_$i
AGT_MS_PRIO=low run low_$i
done
