#!/bin/bash
# round 5, GPU session O: same-box A/B of the round-4 tree (r04tree/, not tracked) against this round's build: c5, c2, c2k20, c4, c3, c3pairs
set -o pipefail
OUT=gpurun_out/r5o
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
run c5_new_$i . python3 bench.py --workload c5 --no-cpu-baseline
run c5_old_$i r04tree python3 bench.py --workload c5 --no-cpu-baseline
done
run c2k20_new . python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras
run c2k20_old r04tree python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras
run c2_new . python3 bench.py --no-cpu-baseline --no-extras
run c2_old r04tree python3 bench.py --no-cpu-baseline --no-extras
run c4_new . python3 bench.py --workload c4 --no-cpu-baseline --no-extras
run c4_old r04tree python3 bench.py --workload c4 --no-cpu-baseline --no-extras
run c3_new . python3 bench.py --workload c3 --steps 256 --warmup 16 --render-frames 8 --no-cpu-baseline --no-extras
run c3_old r04tree python3 bench.py --workload c3 --steps 256 --warmup 16 --render-frames 8 --no-cpu-baseline --no-extras
run pairs_new . python3 bench.py --workload c3pairs --steps 256 --no-cpu-baseline
run pairs_old r04tree python3 bench.py --workload c3pairs --steps 256 --no-cpu-baseline
run c2k20_new2 . python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras
run c2k20_old2 r04tree python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras
