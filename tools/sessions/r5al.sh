#!/bin/bash
# round 5, GPU session AL: stand-alone pose solver register-allocated for 2 / 3 waves per SIMD (easier to place beside tracker / pyramid waves; spills)
set -o pipefail
OUT=gpurun_out/r5al
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
show() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["timing"]["ms_per_step_p10"], d.get("roofline",{}).get("call_spans_us_serial_pass"))
except Exception as e: print("parse", e)
PY
}
run() { name=$1; lib=$2; shift 2; echo "== $name"; AGT_LIB=$lib timeout -k 10 300 python3 tools/knobbench.py --workload c3pairs --steps 1024 --no-cpu-baseline "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "rc=$?"; show "$OUT/$name.json"; }
for i in 1 2; do
run knobs_$i libagt_hip_knobs.so
run occ2_$i libagt_hip_exp_pnpocc2.so
run occ3_$i libagt_hip_exp_pnpocc3.so
done
