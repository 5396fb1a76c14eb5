#!/bin/bash
# round 5, GPU session E: lean horizontal / vertical passes (hgroup8b / vgroup8b) in the rolling pyramid kernels: bit-exactness and
# the pipelined cold-pair step; float-order tests with the accepted / rejected split
set -o pipefail
OUT=gpurun_out/r5e
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
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x > "$OUT/parity_prod.log" 2>&1; echo "parity prod rc=$?"; tail -3 "$OUT/parity_prod.log"
for oh in 2 8; do
  AGT_TEST_LIB=libagt_hip_knobs.so AGT_PYR4=1 AGT_PYR4_REV=1 AGT_PYR4_OH=$oh timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "pyramid_build_all_levels or pyr_down or lk_bit_exact" > "$OUT/pyr_rev_oh$oh.log" 2>&1; echo "pyr rev oh$oh rc=$?"; tail -3 "$OUT/pyr_rev_oh$oh.log"
  AGT_TEST_LIB=libagt_hip_knobs.so AGT_PYR4=1 AGT_PYR4_OH=$oh timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "pyramid_build_all_levels or pyr_down or lk_bit_exact" > "$OUT/pyr_fwd_oh$oh.log" 2>&1; echo "pyr fwd oh$oh rc=$?"; tail -3 "$OUT/pyr_fwd_oh$oh.log"
done
run prod python3 bench.py $P
AGT_PYR4=1 AGT_PYR4_REV=1 AGT_PYR4_OH=8 run p4rev_oh8 python3 tools/knobbench.py $P
AGT_PYR4=1 AGT_PYR4_REV=1 AGT_PYR4_OH=6 run p4rev_oh6 python3 tools/knobbench.py $P
AGT_PYR4=1 AGT_PYR4_REV=1 AGT_PYR4_OH=10 run p4rev_oh10 python3 tools/knobbench.py $P
AGT_PYR4=1 AGT_PYR4_REV=1 AGT_PYR4_OH=12 run p4rev_oh12 python3 tools/knobbench.py $P
AGT_PYR4=1 AGT_PYR4_OH=8 run p4fwd_oh8 python3 tools/knobbench.py $P
run prod2 python3 bench.py $P
AGT_PYR4=1 AGT_PYR4_REV=1 AGT_PYR4_OH=8 run p4rev_oh8_b python3 tools/knobbench.py $P
AGT_LIB=libagt_hip_knobs.so AGT_PYR4=1 AGT_PYR4_REV=1 AGT_PYR4_OH=8 PAIRS_EXP=nolk run p4rev_nolk python3 tools/pairs_exp.py $P
PAIRS_EXP=nolk run prod_nolk python3 tools/pairs_exp.py $P
run c3_prod python3 bench.py --workload c3 --steps 256 --warmup 16 --render-frames 8 --no-cpu-baseline --no-extras
run c2k20 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras
