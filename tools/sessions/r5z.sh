#!/bin/bash
# round 5, GPU session Z: where the 0.5 % of the 20-step c2 blocks went (product / knobs / knobs with MachineLICM on agt_step.hip / round-4 tree)
set -o pipefail
OUT=gpurun_out/r5z
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
show() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["timing"]["ms_per_step_p10"])
except Exception as e: print("parse", e)
PY
}
run() { name=$1; dir=$2; shift 2; echo "== $name"; (cd $dir && timeout -k 10 400 "$@") > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "rc=$?"; show "$OUT/$name.json"; }
A="--steps 20 --warmup 5 --blocks 45 --no-cpu-baseline --no-extras"
for i in 1 2 3; do
run new_$i . python3 bench.py $A
run old_$i r04tree python3 bench.py $A
AGT_LIB=libagt_hip_knobs.so run knobs_$i . python3 tools/knobbench.py $A
AGT_LIB=libagt_hip_exp_licm_step.so run licm_$i . python3 tools/knobbench.py $A
done
