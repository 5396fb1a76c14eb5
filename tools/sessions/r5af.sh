#!/bin/bash
# round 5, GPU session AF: where the HIP runtime puts kernel arguments (HIP_FORCE_DEV_KERNARG unset / 0 / 1)
set -o pipefail
OUT=gpurun_out/r5af
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
show() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["timing"]["ms_per_step_p10"])
except Exception as e: print("parse", e)
PY
}
run() { name=$1; shift; echo "== $name"; timeout -k 10 400 "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "rc=$?"; show "$OUT/$name.json"; }
for i in 1 2; do
for k in unset 0 1; do
if [ $k = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$k; fi
run c2k20_k${k}_$i python3 bench.py --steps 20 --warmup 5 --blocks 45 --no-cpu-baseline --no-extras
run c2_k${k}_$i python3 bench.py --no-cpu-baseline --no-extras
run c4_k${k}_$i python3 bench.py --workload c4 --no-cpu-baseline --no-extras
run c5_k${k}_$i python3 bench.py --workload c5 --no-cpu-baseline
run c3_k${k}_$i python3 bench.py --workload c3 --steps 256 --warmup 16 --render-frames 8 --no-cpu-baseline
run c3pairs_k${k}_$i python3 bench.py --workload c3pairs --steps 256 --no-cpu-baseline
done
done
