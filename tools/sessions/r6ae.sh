#!/bin/bash
# round 6, GPU session AE: the one-wave LK kernel with a 136-register allocation (a clobber of v135: at most THREE of its waves on a SIMD, so
# 3,072 corners sit 3 / 3 / 3 / 3 on every CU instead of wherever the CU's wave dispatch puts them) against the shipped 110 (four fit), c3, same box
set -o pipefail
OUT=gpurun_out/r6ae
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
show() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]; print(d["ms_per_step"], d["timing"]["ms_per_step_p10"], r["whole_step"]["frac_of_8TBs"], r["avg_launch_us"], d.get("accepted_frac"))
except Exception as e: print("parse", e)
PY
}
run() { name=$1; lib=$2; shift 2; echo "== $name"; AGT_LIB=$lib timeout -k 10 300 python3 tools/knobbench.py --no-cpu-baseline --workload c3 --steps 600 "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "rc=$?"; show "$OUT/$name.json"; }
for i in 1 2; do
for cu in -1 0 12; do
run base_cu${cu}_$i libagt_hip_knobs.so --stream-lk-cu $cu
run v136_cu${cu}_$i libagt_hip_exp_v136.so --stream-lk-cu $cu
done
done
