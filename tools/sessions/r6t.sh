#!/bin/bash
# round 6, GPU session T: agt_lk_occupancy_cu (exact residency cap in workgroups per CU; the split pipeline's own choice 10): cap parity
# test, c3 with the library's choice against no cap, cold pairs with 7..11 per CU (round 5's "two per SIMD" was 9 after the margin change)
set -o pipefail
OUT=gpurun_out/r6t
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "occupancy or abi or version" > "$OUT/pytest.log" 2>&1; rc=$?; tail -3 "$OUT/pytest.log"; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit $rc
show() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]; print(d["ms_per_step"], d["timing"]["ms_per_step_p10"], r["whole_step"]["frac_of_8TBs"], r["avg_launch_us"], d.get("accepted_frac"))
except Exception as e: print("parse", e)
PY
}
run() { name=$1; shift; echo "== $name"; timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 256 "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "rc=$?"; show "$OUT/$name.json"; }
for i in 1 2; do
run c3_auto_$i --workload c3
run c3_none_$i --workload c3 --stream-lk-cu 0
for cu in 7 8 9 10 11; do run pairs_cu${cu}_$i --workload c3pairs --lk-cu $cu; done
done
