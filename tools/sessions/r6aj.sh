#!/bin/bash
# round 6, GPU session AJ: issue priority of the one-wave LK waves (0 = default / 1 / 2; experiment builds) x strip height of the pyramid role (16 = the
# plan's cap / 4), c3 in 600-step blocks and cold pairs, same box.  (Short strips helped c3: are young pyramid waves just waves the oldest-first arbiter ranks behind the trackers?)
set -o pipefail
OUT=gpurun_out/r6aj
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
show() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]; print(d["ms_per_step"], d["timing"]["ms_per_step_p10"], r["whole_step"]["frac_of_8TBs"], r["avg_launch_us"])
except Exception as e: print("parse", e)
PY
}
run() { name=$1; lib=$2; shift 2; echo "== $name"; AGT_LIB=$lib timeout -k 10 400 python3 tools/knobbench.py --no-cpu-baseline "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "rc=$?"; show "$OUT/$name.json"; }
for i in 1 2; do
for v in knobs exp_lkp1 exp_lkp2; do
for oh in 0 4; do
AGT_PYR4_OH=$oh run c3_${v}_oh${oh}_$i libagt_hip_$v.so --workload c3 --steps 600
done
AGT_PYR4_OH=0 run pairs_${v}_$i libagt_hip_$v.so --workload c3pairs --steps 1024 --warmup 32
done
done
