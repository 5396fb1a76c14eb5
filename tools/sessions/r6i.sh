#!/bin/bash
# round 6, GPU session I: search margin of the one-wave LK bodies 9 / 7 / 5 px, same box (knobs builds)
set -o pipefail
OUT=gpurun_out/r6i
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
show() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["timing"]["ms_per_step_p10"], d.get("accepted_frac"))
except Exception as e: print("parse", e)
PY
}
run() { name=$1; lib=$2; shift 2; echo "== $name"; AGT_LIB=$lib timeout -k 10 400 python3 tools/knobbench.py --no-cpu-baseline --no-extras "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "rc=$?"; show "$OUT/$name.json"; }
for i in 1 2; do
for v in m9 m7 m5; do
run c3_${v}_$i libagt_hip_exp_$v.so --workload c3 --steps 256 --warmup 16 --render-frames 8
run c3pairs_${v}_$i libagt_hip_exp_$v.so --workload c3pairs --steps 256
done
done
