#!/bin/bash
# round 5, GPU session W: max-ILP scheduling per file (pose kernels with the AMDGPU pressure trackers, LK, dense), same-box A/B
set -o pipefail
OUT=gpurun_out/r5w
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
for v in knobs exp_ilp exp_ilpt exp_ilpt_step; do
run c2k20_${v}_$i libagt_hip_$v.so --steps 20 --warmup 5
run c2_${v}_$i libagt_hip_$v.so
run c4_${v}_$i libagt_hip_$v.so --workload c4
done
for v in knobs exp_ilpt exp_ilp_dense; do
run c5_${v}_$i libagt_hip_$v.so --workload c5
done
for v in knobs exp_ilpt exp_ilp_lk; do
run c3_${v}_$i libagt_hip_$v.so --workload c3 --steps 256 --warmup 16 --render-frames 8
run c3pairs_${v}_$i libagt_hip_$v.so --workload c3pairs --steps 256
done
done
