#!/bin/bash
# round 6, GPU session B: the one-wave LK iteration diet (dot2 taps with int16 weights, scalar weight extraction, scalar window address / box,
# position formed behind the loop): GPU suite once, then c3 / c3pairs / c2 step times, two runs each
set -o pipefail
OUT=gpurun_out/${1:-r6b}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1; rc=$?; tail -5 "$OUT/pytest.log"; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit $rc
show() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["timing"]["ms_per_step_p10"], d.get("accepted_frac"), d.get("roofline",{}).get("separate_kernel_spans_us"))
except Exception as e: print("parse", e)
PY
}
run() { name=$1; shift; echo "== $name"; timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-extras "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; r=$?; echo "rc=$r"; show "$OUT/$name.json"; return $r; }
for i in 1 2; do
run c3_$i --workload c3 --steps 256 --warmup 16 --render-frames 8 && run c3pairs_$i --workload c3pairs --steps 256 && run c2k20_$i --steps 20 --warmup 5 --blocks 45 || exit 1
done
