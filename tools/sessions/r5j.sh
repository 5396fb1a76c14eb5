#!/bin/bash
# round 5, GPU session J: whole suite on the current build; cold pairs with the LK occupancy cap as a library option; pose-solver stamps
set -o pipefail
OUT=gpurun_out/r5j
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
timeout -k 10 1100 python3 -m pytest tests -q -m gpu -x > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest.log"
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
run pairs_ctx5 python3 bench.py $P --pair-contexts 5
run pairs_ctx3 python3 bench.py $P --pair-contexts 3
timeout -k 10 300 python3 tools/pnpstamps.py 12 > "$OUT/pnpstamps12.txt" 2>&1; tail -12 "$OUT/pnpstamps12.txt"
for B in 2 8; do
  AGT_LIB=libagt_hip_knobs.so AGT_PNP_COOP_GROUP=0 timeout -k 10 300 python3 tools/coop240.py $B 16 2>&1 | tail -1 | sed 's/^/perframe /'
  AGT_LIB=libagt_hip_knobs.so AGT_PNP_COOP_GROUP=1 timeout -k 10 300 python3 tools/coop240.py $B 16 2>&1 | tail -1 | sed 's/^/group    /'
  timeout -k 10 300 python3 tools/coop240.py $B 16 2>&1 | tail -1 | sed 's/^/product  /'
done
run c5 python3 bench.py --workload c5 --no-cpu-baseline
