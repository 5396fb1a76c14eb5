#!/bin/bash
# round 5, GPU session S: c3 (64 streams) with the LK occupancy cap of the cold-pair bench
set -o pipefail
OUT=gpurun_out/r5s
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
show() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d.get("accepted_frac"))
except Exception as e: print("parse", e)
PY
}
run() { name=$1; shift; echo "== $name"; timeout -k 10 300 python3 bench.py --workload c3 --steps 256 --warmup 16 --render-frames 8 --no-cpu-baseline "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "rc=$?"; show "$OUT/$name.json"; }
run occ0_a
run occ2_a --stream-lk-occupancy 2
run occ3_a --stream-lk-occupancy 3
run occ4_a --stream-lk-occupancy 4
run occ0_b
run occ2_b --stream-lk-occupancy 2
run occ3_b --stream-lk-occupancy 3
