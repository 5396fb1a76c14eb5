#!/bin/bash
# round 5, GPU session AK: ... with more hardware queues (GPU_MAX_HW_QUEUES)
set -o pipefail
OUT=gpurun_out/r5ak
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
show() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["timing"]["ms_per_step_p10"])
except Exception as e: print("parse", e)
PY
}
run() { name=$1; shift; echo "== $name"; timeout -k 10 300 python3 bench.py --workload c3pairs --steps 1024 --no-cpu-baseline "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "rc=$?"; show "$OUT/$name.json"; }
run same4_q4
for q in 5 6 8; do
export GPU_MAX_HW_QUEUES=$q
run same4_q$q
run high4_q$q --pnp-stream high
done
