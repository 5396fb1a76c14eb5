#!/bin/bash
# round 5, GPU session L: cold pairs with the occupancy cap really applied; c3 streams at other group depths
set -o pipefail
OUT=gpurun_out/r5l
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "occupancy or lk_large_batch" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest.log"
show() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d.get("roofline",{}).get("frac"), d.get("roofline",{}).get("whole_step",{}).get("frac_of_8TBs"), d.get("roofline",{}).get("call_spans_us_serial_pass"))
except Exception as e: print("parse", e)
PY
}
run() { name=$1; shift; echo "== $name"; timeout -k 10 400 "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "rc=$?"; show "$OUT/$name.json"; }
P="--workload c3pairs --steps 256 --no-cpu-baseline"
run pairs python3 bench.py $P
run pairs_cap0 python3 bench.py $P --lk-occupancy 0
run pairs_cap1 python3 bench.py $P --lk-occupancy 1
run pairs_cap3 python3 bench.py $P --lk-occupancy 3
run pairs_b python3 bench.py $P
run pairs_cap0_b python3 bench.py $P --lk-occupancy 0
C3="--workload c3 --steps 256 --warmup 16 --render-frames 8 --no-cpu-baseline --no-extras"
run c3_d16 python3 bench.py $C3
run c3_d32 python3 bench.py $C3 --depth 32
run c3_d24 python3 bench.py $C3 --depth 24
run c3_d8 python3 bench.py $C3 --depth 8
run c3_d16b python3 bench.py $C3
