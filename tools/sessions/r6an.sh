#!/bin/bash
# round 6, GPU session AN: with the pyramid role in strips of 6 -- frames per group (8 / 16 / 24 / 32) and LK residency cap (8 / 10 / 12 / none) again, c3, same box
set -o pipefail
OUT=gpurun_out/r6an
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
show() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]; print(d["ms_per_step"], d["timing"]["ms_per_step_p10"], r["whole_step"]["frac_of_8TBs"], d.get("accepted_frac"))
except Exception as e: print("parse", e)
PY
}
run() { name=$1; shift; echo "== $name"; timeout -k 10 300 python3 bench.py --no-cpu-baseline --workload c3 --steps 600 "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "rc=$?"; show "$OUT/$name.json"; }
for i in 1 2; do
run d16_auto_$i
run d8_$i --depth 8
run d24_$i --depth 24
run d32_$i --depth 32
run cu8_$i --stream-lk-cu 8
run cu12_$i --stream-lk-cu 12
run cu0_$i --stream-lk-cu 0
run cu9_$i --stream-lk-cu 9
run cu11_$i --stream-lk-cu 11
done
