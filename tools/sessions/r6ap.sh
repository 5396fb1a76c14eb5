#!/bin/bash
# round 6, GPU session AP: cold pairs with the co-tenant priority on -- residency cap 7 / 8 / 9 / 10 again, and the pair launch in strips of 14, same box
set -o pipefail
OUT=gpurun_out/r6ap
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
show() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]; print(d["ms_per_step"], d["timing"]["ms_per_step_p10"], r["whole_step"]["frac_of_8TBs"], r["avg_launch_us"])
except Exception as e: print("parse", e)
PY
}
run() { name=$1; shift; echo "== $name"; timeout -k 10 400 python3 tools/knobbench.py --no-cpu-baseline --workload c3pairs --steps 1024 --warmup 32 "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "rc=$?"; show "$OUT/$name.json"; }
for i in 1 2; do
for cu in 8 7 9 10; do run cu${cu}_$i --lk-cu $cu; done
AGT_PYR4_OH=14 run cu8_oh14_$i --lk-cu 8
AGT_PYR4_OH=12 run cu8_oh12_$i --lk-cu 8
done
