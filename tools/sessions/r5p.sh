#!/bin/bash
# round 5, GPU session P: which change costs c4 / c5 / c2k20 their 0.5-1.8 %: step-kernel variants on one box
set -o pipefail
OUT=gpurun_out/r5p
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
show() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])
except Exception as e: print("parse", e)
PY
}
run() { name=$1; dir=$2; shift 2; echo "== $name"; (cd $dir && timeout -k 10 400 "$@") > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "rc=$?"; show "$OUT/$name.json"; }
for i in 1 2; do
for v in knobs exp_xs3 exp_xs3seq; do
  AGT_LIB=libagt_hip_$v.so run c4_${v}_$i . python3 tools/knobbench.py --workload c4 --no-cpu-baseline --no-extras
done
run c4_old_$i r04tree python3 bench.py --workload c4 --no-cpu-baseline --no-extras
done
for v in knobs exp_xs3 exp_xs3seq; do
  AGT_LIB=libagt_hip_$v.so run c5_$v . python3 tools/knobbench.py --workload c5 --no-cpu-baseline
  AGT_LIB=libagt_hip_$v.so run c2k20_$v . python3 tools/knobbench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras
done
run c5_old r04tree python3 bench.py --workload c5 --no-cpu-baseline
run c2k20_old r04tree python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras
run c5_prod . python3 bench.py --workload c5 --no-cpu-baseline
run c4_prod . python3 bench.py --workload c4 --no-cpu-baseline --no-extras
