#!/bin/bash
# round 5, GPU session K: cold pairs with the occupancy cap (fixed LDS size); dynamic instruction counters of one pose solve
set -o pipefail
OUT=gpurun_out/r5k
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
for c in SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY; do
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pnp_$c" -- python3 tools/pnpone.py > "$OUT/pnp_$c.stdout" 2> "$OUT/pnp_$c.stderr"; echo "pmc $c rc=$?"
done
tail -2 "$OUT/pnp_SQ_INSTS_VALU.stdout"
