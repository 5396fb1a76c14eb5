#!/bin/bash
# round 5, GPU session X: max-ILP scheduling of agt_step.hip with / without MachineLICM, same-box A/B
set -o pipefail
OUT=gpurun_out/r5x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
show() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])
except Exception as e: print("parse", e)
PY
}
run() { name=$1; lib=$2; shift 2; echo "== $name"; AGT_LIB=$lib timeout -k 10 400 python3 tools/knobbench.py "$@" --no-cpu-baseline --no-extras > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "rc=$?"; show "$OUT/$name.json"; }
for i in 1 2; do
for v in knobs exp_ilp exp_ilp_step exp_ilp_licm exp_ilp_licm_pnp; do
run c2k20_${v}_$i libagt_hip_$v.so --steps 20 --warmup 5
run c2_${v}_$i libagt_hip_$v.so
run c4_${v}_$i libagt_hip_$v.so --workload c4
done
for v in knobs exp_ilp_step exp_ilp_licm exp_ilp_licm_pnp; do
run c3_${v}_$i libagt_hip_$v.so --workload c3 --steps 256 --warmup 16 --render-frames 8
done
done
for v in knobs exp_ilp_step exp_ilp_licm; do
for B in 2 8; do
echo "== coop240_${v}_B$B"; AGT_LIB=libagt_hip_$v.so timeout -k 10 300 python3 tools/coop240.py $B 16 > "$OUT/coop240_${v}_B$B.txt" 2>&1; echo "rc=$?"; tail -1 "$OUT/coop240_${v}_B$B.txt"
done
done
