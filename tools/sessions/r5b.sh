#!/bin/bash
# round 5, GPU session B: what bounds the pipelined cold-pair step -- ablations (tools/pairs_exp.py), more contexts / hardware queues,
# HBM-side traffic of the two-level rolling pass inside the pipeline
set -o pipefail
OUT=gpurun_out/r5b
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
show() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d.get("roofline",{}).get("call_spans_us_serial_pass"))
except Exception as e: print("parse", e)
PY
}
run() { name=$1; shift; echo "== $name"; timeout -k 10 300 "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "rc=$?"; show "$OUT/$name.json"; }
P="--workload c3pairs --steps 256 --no-cpu-baseline"
run base python3 tools/pairs_exp.py $P
PAIRS_EXP=lk1 run lk1 python3 tools/pairs_exp.py $P
PAIRS_EXP=nolk run nolk python3 tools/pairs_exp.py $P
PAIRS_EXP=nopnp run nopnp python3 tools/pairs_exp.py $P
PAIRS_EXP=nonext run nonext python3 tools/pairs_exp.py $P
PAIRS_EXP=nopyr run nopyr python3 tools/pairs_exp.py $P
GPU_MAX_HW_QUEUES=8 run q8_ctx4 python3 bench.py $P
GPU_MAX_HW_QUEUES=8 run q8_ctx6 python3 bench.py $P --pair-contexts 6
GPU_MAX_HW_QUEUES=8 run q8_ctx8 python3 bench.py $P --pair-contexts 8
run ctx6 python3 bench.py $P --pair-contexts 6
run ctx8 python3 bench.py $P --pair-contexts 8
export AGT_LIB=libagt_hip_knobs.so AGT_PYR4=1 AGT_PYR4_OH=8
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pyr4_$c" -- python3 tools/knobbench.py --workload c3pairs --steps 24 --warmup 8 --blocks 2 --pair-contexts 1 --no-cpu-baseline > "$OUT/pyr4_$c.stdout" 2> "$OUT/pyr4_$c.stderr"; echo "pmc $c rc=$?"
done
PAIRS_EXP=lk1 run pyr4_lk1 python3 tools/pairs_exp.py $P
PAIRS_EXP=nolk run pyr4_nolk python3 tools/pairs_exp.py $P
run pyr4_base python3 tools/pairs_exp.py $P
du -sh "$OUT"
