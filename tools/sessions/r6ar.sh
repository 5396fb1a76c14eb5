#!/bin/bash
# round 6, GPU session AR: the pyramid role's strips of 6 at other stream counts of the split pipeline (8 / 16 / 32 / 128 streams): shipped against
# AGT_PYR4_OH=16 (the old cap), knobs build, same box
set -o pipefail
OUT=gpurun_out/r6ar
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
show() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["timing"]["ms_per_step_p10"], d.get("accepted_frac"))
except Exception as e: print("parse", e)
PY
}
run() { name=$1; shift; echo "== $name"; timeout -k 10 300 python3 tools/knobbench.py --no-cpu-baseline --workload c3 --steps 256 --render-frames 8 "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "rc=$?"; show "$OUT/$name.json"; }
for B in 8 16 32 128; do
for i in 1 2; do
run b${B}_oh6_$i --streams $B
AGT_PYR4_OH=16 run b${B}_oh16_$i --streams $B
done
done
