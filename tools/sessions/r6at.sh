#!/bin/bash
# round 6, GPU session AT: the rolling two-level pass with 16 rows in flight per lane (experiment build -DAGT_PYR4_RING=16) against 8: the pass alone
# (pyrbench), cold pairs, c3; same box
set -o pipefail
OUT=gpurun_out/r6at
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
for lib in libagt_hip_knobs.so libagt_hip_exp_ring16.so; do echo "== pyrbench $lib"; AGT_LIB=$lib timeout -k 10 200 python3 tools/pyrbench.py 60 2>&1 | tail -2; done
show() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]; print(d["ms_per_step"], d["timing"]["ms_per_step_p10"], r["whole_step"]["frac_of_8TBs"], r["avg_launch_us"], (r.get("alone") or {}).get("avg_launch_us"))
except Exception as e: print("parse", e)
PY
}
run() { name=$1; lib=$2; shift 2; echo "== $name"; AGT_LIB=$lib timeout -k 10 400 python3 tools/knobbench.py --no-cpu-baseline "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "rc=$?"; show "$OUT/$name.json"; }
for i in 1 2; do
for v in knobs exp_ring16; do
run pairs_${v}_$i libagt_hip_$v.so --workload c3pairs --steps 1024 --warmup 32
run c3_${v}_$i libagt_hip_$v.so --workload c3 --steps 600
done
done
