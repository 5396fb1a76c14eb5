#!/bin/bash
# round 5, GPU session A: FETCH_SIZE calibration for dword tile loads; c3pairs baseline / two-level rolling pass inside the 4-context
# pipeline (knobs build); kernel trace with timestamps of the pipelined c3pairs run
set -o pipefail
OUT=gpurun_out/r5a
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
rocprofv3 -L > "$OUT/counters_avail.txt" 2>&1
./tools/fetchcal 1 > "$OUT/fetchcal_expected.json" 2> "$OUT/fetchcal.err" || exit 1
for c in FETCH_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_MISS_sum TCC_HIT_sum TCC_REQ_sum; do
  timeout -k 10 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/cal_$c" -- ./tools/fetchcal 3 > "$OUT/cal_$c.stdout" 2> "$OUT/cal_$c.stderr"; echo "cal $c rc=$?"
done
run() { name=$1; shift; echo "== $name"; timeout -k 10 300 "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "rc=$?"; tail -c 600 "$OUT/$name.json" | head -c 0; python3 - "$OUT/$name.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d.get("roofline",{}).get("whole_step"), d.get("roofline",{}).get("call_spans_us_serial_pass"))
except Exception as e: print("parse", e)
PY
}
P="--workload c3pairs --steps 256 --no-cpu-baseline"
run pairs_prod_1 python3 bench.py $P
run pairs_knobs_ctl python3 tools/knobbench.py $P
AGT_PYR4=1 AGT_PYR4_OH=4 run pairs_pyr4_oh4 python3 tools/knobbench.py $P
AGT_PYR4=1 AGT_PYR4_OH=8 run pairs_pyr4_oh8 python3 tools/knobbench.py $P
AGT_PYR4=1 AGT_PYR4_OH=16 run pairs_pyr4_oh16 python3 tools/knobbench.py $P
run pairs_prod_2 python3 bench.py $P
run pairs_prod_ctx1 python3 bench.py $P --pair-contexts 1
run pairs_prod_ctx2 python3 bench.py $P --pair-contexts 2
run pairs_prod_ctx3 python3 bench.py $P --pair-contexts 3
run c3_prod python3 bench.py --workload c3 --steps 256 --warmup 16 --render-frames 8 --no-cpu-baseline --no-extras
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/pairs_trace" -- python3 bench.py --workload c3pairs --steps 64 --warmup 8 --blocks 3 --no-cpu-baseline > "$OUT/pairs_trace.stdout" 2> "$OUT/pairs_trace.stderr"; echo "trace rc=$?"
find "$OUT" -name "*.csv" | head -50
du -sh "$OUT"
