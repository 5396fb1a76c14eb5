#!/bin/bash
# round 5, GPU session AZ: uneven halves of the 64-stream LK chains (AGT_SPLIT_B1, knobs build): 1,536 + 1,536 corners put 1.5 waves per SIMD per launch
set -o pipefail
OUT=gpurun_out/r5az
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
show() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["timing"]["ms_per_step_p10"], d.get("accepted_frac"))
except Exception as e: print("parse", e)
PY
}
run() { name=$1; shift; echo "== $name"; AGT_LIB=libagt_hip_knobs.so timeout -k 10 300 python3 tools/knobbench.py --workload c3 --steps 256 --warmup 16 --render-frames 8 --no-cpu-baseline "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "rc=$?"; show "$OUT/$name.json"; }
for i in 1 2; do
unset AGT_SPLIT_B1; run b32_$i
for b in 21 26 38 43 48; do export AGT_SPLIT_B1=$b; run b${b}_$i; done
done
