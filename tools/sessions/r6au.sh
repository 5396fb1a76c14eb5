#!/bin/bash
# round 6, GPU session AU: issue priority of the stand-alone pose kernel (0 shipped / 2 / 3) now that co-tenant trackers run at 1: cold pairs, same box
set -o pipefail
OUT=gpurun_out/r6au
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
show() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]; print(d["ms_per_step"], d["timing"]["ms_per_step_p10"], r["whole_step"]["frac_of_8TBs"], r["call_spans_us_pipelined_pass"])
except Exception as e: print("parse", e)
PY
}
run() { name=$1; lib=$2; shift 2; echo "== $name"; AGT_LIB=$lib timeout -k 10 400 python3 tools/knobbench.py --no-cpu-baseline --workload c3pairs --steps 1024 --warmup 32 "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "rc=$?"; show "$OUT/$name.json"; }
for i in 1 2 3; do
run p0_$i libagt_hip_knobs.so
run p2_$i libagt_hip_exp_pnp2.so
run p3_$i libagt_hip_exp_pnp3.so
done
