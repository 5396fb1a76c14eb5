#!/bin/bash
# round 6, GPU session AK: shipped forms of session AJ's findings -- the split pipeline's pyramid role in strips of 4 level-2 rows, tracker waves of
# co-tenant contexts at issue priority 1: the GPU suite, then c3 (600 / 256-step blocks; AGT_PYR4_OH=16 in the knobs build = the old strips), cold pairs, c2 blocks of 20, c4, c5
set -o pipefail
OUT=gpurun_out/r6ak
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1; rc=$?; tail -4 "$OUT/pytest.log"; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit $rc
show() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]; print(d["value"], d["ms_per_step"], d["timing"]["ms_per_step_p10"], (r.get("whole_step") or {}).get("frac_of_8TBs"), r["avg_launch_us"], d.get("accepted_frac"))
except Exception as e: print("parse", e)
PY
}
run() { name=$1; shift; echo "== $name"; timeout -k 10 400 python3 "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "rc=$?"; show "$OUT/$name.json"; }
for i in 1 2; do
run c3_600_$i bench.py --no-cpu-baseline --workload c3 --steps 600
AGT_PYR4_OH=16 run c3_600_knobs_oh16_$i tools/knobbench.py --no-cpu-baseline --workload c3 --steps 600
run c3_256_$i bench.py --no-cpu-baseline --workload c3 --steps 256
run pairs_$i bench.py --no-cpu-baseline --workload c3pairs --steps 1024 --warmup 32
done
run c2k20 bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5
run c4 bench.py --no-cpu-baseline --no-extras --workload c4
run c5 bench.py --no-cpu-baseline --workload c5
