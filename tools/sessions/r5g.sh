#!/bin/bash
# round 5, GPU session G: new GPU tests (RCCL world 1, device info, float orders); issue-priority / LK-occupancy experiments on the cold-pair step
set -o pipefail
OUT=gpurun_out/r5g
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
timeout -k 10 900 python3 -m pytest tests/test_gpu_distributed.py tests/test_gpu_float_order.py -q -s -m gpu > "$OUT/newtests.log" 2>&1; echo "newtests rc=$?"; grep "deviation-1\|passed\|failed\|Error\|error" "$OUT/newtests.log" | head -40
show() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d.get("roofline",{}).get("frac"), d.get("roofline",{}).get("call_spans_us_serial_pass"))
except Exception as e: print("parse", e)
PY
}
run() { name=$1; shift; echo "== $name"; timeout -k 10 400 "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "rc=$?"; show "$OUT/$name.json"; }
P="--workload c3pairs --steps 256 --no-cpu-baseline"
run base python3 tools/knobbench.py $P
AGT_LIB=libagt_hip_exp_pyrprio.so run pyrprio python3 tools/knobbench.py $P
AGT_LIB=libagt_hip_exp_pnpprio.so run pnpprio python3 tools/knobbench.py $P
AGT_LIB=libagt_hip_exp_bothprio.so run bothprio python3 tools/knobbench.py $P
AGT_LK_LDS_PAD=4096 run lkpad4k python3 tools/knobbench.py $P
AGT_LK_LDS_PAD=10240 run lkpad10k python3 tools/knobbench.py $P
run base2 python3 tools/knobbench.py $P
AGT_LIB=libagt_hip_exp_pyrprio.so run pyrprio2 python3 tools/knobbench.py $P
AGT_LIB=libagt_hip_exp_pnpprio.so run pnpprio2 python3 tools/knobbench.py $P
run base3 python3 tools/knobbench.py $P
