#!/bin/bash
# round 5, GPU session C: pyramid waves beside the LK waves -- ring depth / register footprint variants in the pipelined cold-pair
# step; first run of tests/test_gpu_float_order.py
set -o pipefail
OUT=gpurun_out/r5c
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
timeout -k 10 900 python3 -m pytest tests/test_gpu_float_order.py -x -q -s -m gpu > "$OUT/float_order.log" 2>&1; echo "float_order rc=$?"; tail -25 "$OUT/float_order.log"
run base python3 tools/knobbench.py $P
AGT_LIB=libagt_hip_exp_p3r4.so run p3r4 python3 tools/knobbench.py $P
AGT_LIB=libagt_hip_exp_p3r4v64.so run p3r4v64 python3 tools/knobbench.py $P
AGT_LIB=libagt_hip_exp_p3r4v64.so AGT_PYR3_OH=8 run p3r4v64_oh8 python3 tools/knobbench.py $P
AGT_LIB=libagt_hip_exp_p3r16.so AGT_PYR3_OH=16 run p3r16_oh16 python3 tools/knobbench.py $P
AGT_LIB=libagt_hip_exp_p3r16.so AGT_PYR3_OH=8 run p3r16_oh8 python3 tools/knobbench.py $P
AGT_PYR3_OH=8 run p3r8_oh8 python3 tools/knobbench.py $P
AGT_LIB=libagt_hip_exp_p4r16.so AGT_PYR4=1 AGT_PYR4_OH=8 run p4r16_oh8 python3 tools/knobbench.py $P
AGT_LIB=libagt_hip_exp_p4r16.so AGT_PYR4=1 AGT_PYR4_OH=16 run p4r16_oh16 python3 tools/knobbench.py $P
AGT_LIB=libagt_hip_exp_p4r16.so AGT_PYR4=1 AGT_PYR4_OH=12 run p4r16_oh12 python3 tools/knobbench.py $P
run base2 python3 tools/knobbench.py $P
