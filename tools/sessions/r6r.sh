#!/bin/bash
# round 6, GPU session R: occupancy cap of the one-wave LK launches after the diet: c3 (split pipeline, one context: 3,072 corners = 3 waves
# on every SIMD if spread evenly; the corner-life histogram is trimodal -- 17 / 21 / 25 us -- as if SIMDs held 2 / 3 / 4) with caps 0 / 3,
# cold pairs (4 contexts) with caps 2 / 3
set -o pipefail
OUT=gpurun_out/r6r
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
show() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]; print(d["value"], d["ms_per_step"], d["timing"]["ms_per_step_p10"], r["whole_step"]["frac_of_8TBs"], r["frac"], r["avg_launch_us"], d.get("accepted_frac"))
except Exception as e: print("parse", e)
PY
}
run() { name=$1; shift; echo "== $name"; timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 256 "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "rc=$?"; show "$OUT/$name.json"; }
for i in 1 2; do
run c3_cap0_$i --workload c3
run c3_cap3_$i --workload c3 --stream-lk-occupancy 3
run c3_cap2_$i --workload c3 --stream-lk-occupancy 2
run pairs_cap2_$i --workload c3pairs
run pairs_cap3_$i --workload c3pairs --lk-occupancy 3
done
