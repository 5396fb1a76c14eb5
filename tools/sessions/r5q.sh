#!/bin/bash
# round 5, GPU session Q: whole suite + same-box A/B against the round-4 tree after the literal 8-way deal is back in the fused kernels
set -o pipefail
OUT=gpurun_out/r5q
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
timeout -k 10 1100 python3 -m pytest tests -q -m gpu -x > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest.log"
show() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])
except Exception as e: print("parse", e)
PY
}
run() { name=$1; dir=$2; shift 2; echo "== $name"; (cd $dir && timeout -k 10 400 "$@") > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "rc=$?"; show "$OUT/$name.json"; }
for i in 1 2; do
run c4_new_$i . python3 bench.py --workload c4 --no-cpu-baseline --no-extras
run c4_old_$i r04tree python3 bench.py --workload c4 --no-cpu-baseline --no-extras
run c5_new_$i . python3 bench.py --workload c5 --no-cpu-baseline
run c5_old_$i r04tree python3 bench.py --workload c5 --no-cpu-baseline
run c2k20_new_$i . python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras
run c2k20_old_$i r04tree python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras
done
run c2_new . python3 bench.py --no-cpu-baseline --no-extras
run c2_old r04tree python3 bench.py --no-cpu-baseline --no-extras
