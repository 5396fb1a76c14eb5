#!/bin/bash
# round 6, GPU session AA: (1) general LK body against the compiled-in windows (test + timing), (2) cache-policy bits of the rolling pyramid
# pass (nt loads / nt stores / both; experiment builds) on the cold-pair step and c3, same box
set -o pipefail
OUT=gpurun_out/r6aa
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "general_body or any_window" > "$OUT/pytest.log" 2>&1; rc=$?; tail -3 "$OUT/pytest.log"; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 tools/lkanybench.py > "$OUT/lkanybench.txt" 2>&1; echo "lkanybench rc=$?"; cat "$OUT/lkanybench.txt"
show() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]; print(d["ms_per_step"], d["timing"]["ms_per_step_p10"], r["whole_step"]["frac_of_8TBs"], r["avg_launch_us"], (r.get("alone") or {}).get("avg_launch_us"), d.get("accepted_frac"))
except Exception as e: print("parse", e)
PY
}
run() { name=$1; lib=$2; shift 2; echo "== $name"; AGT_LIB=$lib timeout -k 10 400 python3 tools/knobbench.py --no-cpu-baseline "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "rc=$?"; show "$OUT/$name.json"; }
for i in 1 2; do
for v in knobs exp_ldnt exp_stnt exp_bothnt; do
run pairs_${v}_$i libagt_hip_$v.so --workload c3pairs --steps 1024 --warmup 32
run c3_${v}_$i libagt_hip_$v.so --workload c3 --steps 600
done
done
