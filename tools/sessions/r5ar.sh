#!/bin/bash
# round 5, GPU session AR: how the runtime waits for completion signals (HSA_ENABLE_INTERRUPT=0: busy polling) -- driver-style c2 blocks, per-call latencies
set -o pipefail
OUT=gpurun_out/r5ar
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
show() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["timing"]["ms_per_step_p10"], d.get("drop_in_frame_us_median"), d.get("live_frame_us_median"), (d.get("per_call_latency_us") or {}).get("calcOpticalFlowPyrLK_1280x720_N48"))
except Exception as e: print("parse", e)
PY
}
run() { name=$1; shift; echo "== $name"; timeout -k 10 400 "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "rc=$?"; show "$OUT/$name.json"; }
A="--steps 20 --warmup 5 --blocks 45 --no-cpu-baseline"
for i in 1 2; do
unset HSA_ENABLE_INTERRUPT
run c2k20_default_$i python3 bench.py $A
export HSA_ENABLE_INTERRUPT=0
run c2k20_poll_$i python3 bench.py $A
done
unset HSA_ENABLE_INTERRUPT
run c5_default python3 bench.py --workload c5 --no-cpu-baseline
export HSA_ENABLE_INTERRUPT=0
run c5_poll python3 bench.py --workload c5 --no-cpu-baseline
