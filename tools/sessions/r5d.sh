#!/bin/bash
# round 5, GPU session D: alternating strip directions in the two-level rolling pass (AGT_PYR4_REV): bit-exactness, pipelined
# cold-pair step, HBM-side traffic; tests/test_gpu_float_order.py
set -o pipefail
OUT=gpurun_out/r5d
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
timeout -k 10 900 python3 -m pytest tests/test_gpu_float_order.py -q -s -m gpu > "$OUT/float_order.log" 2>&1; echo "float_order rc=$?"; grep "deviation-1\|passed\|failed" "$OUT/float_order.log"
for oh in 2 4 8 16; do
  AGT_TEST_LIB=libagt_hip_knobs.so AGT_PYR4=1 AGT_PYR4_REV=1 AGT_PYR4_OH=$oh timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "pyramid_build_all_levels or pyr_down or lk_bit_exact" > "$OUT/pyr_rev_oh$oh.log" 2>&1; echo "pyr rev oh$oh rc=$?"; tail -3 "$OUT/pyr_rev_oh$oh.log"
done
run base python3 tools/knobbench.py $P
AGT_PYR4=1 AGT_PYR4_OH=8 run p4_oh8 python3 tools/knobbench.py $P
AGT_PYR4=1 AGT_PYR4_REV=1 AGT_PYR4_OH=4 run p4rev_oh4 python3 tools/knobbench.py $P
AGT_PYR4=1 AGT_PYR4_REV=1 AGT_PYR4_OH=6 run p4rev_oh6 python3 tools/knobbench.py $P
AGT_PYR4=1 AGT_PYR4_REV=1 AGT_PYR4_OH=8 run p4rev_oh8 python3 tools/knobbench.py $P
AGT_PYR4=1 AGT_PYR4_REV=1 AGT_PYR4_OH=12 run p4rev_oh12 python3 tools/knobbench.py $P
AGT_PYR4=1 AGT_PYR4_REV=1 AGT_PYR4_OH=16 run p4rev_oh16 python3 tools/knobbench.py $P
run base2 python3 tools/knobbench.py $P
export AGT_LIB=libagt_hip_knobs.so AGT_PYR4=1 AGT_PYR4_REV=1 AGT_PYR4_OH=8
for c in FETCH_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/p4rev_$c" -- python3 tools/knobbench.py --workload c3pairs --steps 24 --warmup 8 --blocks 2 --pair-contexts 1 --no-cpu-baseline > "$OUT/p4rev_$c.stdout" 2> "$OUT/p4rev_$c.stderr"; echo "pmc $c rc=$?"
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/p4rev4_$c" -- python3 tools/knobbench.py --workload c3pairs --steps 64 --warmup 8 --blocks 2 --pair-contexts 4 --no-cpu-baseline > "$OUT/p4rev4_$c.stdout" 2> "$OUT/p4rev4_$c.stderr"; echo "pmc4 $c rc=$?"
done
PAIRS_EXP=nolk run p4rev_nolk python3 tools/pairs_exp.py $P
