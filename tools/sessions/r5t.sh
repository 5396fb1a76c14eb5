#!/bin/bash
# round 5, GPU session T: agt_step.hip (and everything) compiled without MachineLICM -- same-box A/B against the knobs build of the same tree
set -o pipefail
OUT=gpurun_out/r5t
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
for v in knobs exp_nolicm_step exp_nolicm_all; do
run c2k20_${v}_$i libagt_hip_$v.so --steps 20 --warmup 5
run c2_${v}_$i libagt_hip_$v.so
run c4_${v}_$i libagt_hip_$v.so --workload c4
run c5_${v}_$i libagt_hip_$v.so --workload c5
run c3_${v}_$i libagt_hip_$v.so --workload c3 --steps 256 --warmup 16 --render-frames 8
done
done
for v in knobs exp_nolicm_step; do
echo "== coop240 $v"; AGT_LIB=libagt_hip_$v.so timeout -k 10 300 python3 tools/coop240.py > "$OUT/coop240_$v.txt" 2>&1; echo "rc=$?"; tail -5 "$OUT/coop240_$v.txt"
done
AGT_TEST_LIB=libagt_hip_exp_nolicm_step.so timeout -k 10 900 python3 -m pytest tests -q -m gpu -x > "$OUT/pytest_nolicm_step.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest_nolicm_step.log"
