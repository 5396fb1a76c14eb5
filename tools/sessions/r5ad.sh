#!/bin/bash
# round 5, GPU session AD: c4 bisect -- round-4 tree, the tree of commit 0cee8b9 (before the zero-scratch work), HEAD, HEAD with the original thread-index lines
set -o pipefail
OUT=gpurun_out/r5ad
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
show() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["timing"]["ms_per_step_p10"])
except Exception as e: print("parse", e)
PY
}
run() { name=$1; dir=$2; shift 2; echo "== $name"; (cd $dir && timeout -k 10 400 "$@") > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "rc=$?"; show "$OUT/$name.json"; }
for i in 1 2 3; do
run c4_old_$i r04tree python3 bench.py --workload c4 --no-cpu-baseline --no-extras
run c4_r5q_$i r5qtree python3 bench.py --workload c4 --no-cpu-baseline --no-extras
run c4_new_$i . python3 bench.py --workload c4 --no-cpu-baseline --no-extras
AGT_LIB=libagt_hip_knobs.so run c4_knobs_$i . python3 tools/knobbench.py --workload c4 --no-cpu-baseline --no-extras
AGT_LIB=libagt_hip_exp_tidorig.so run c4_tidorig_$i . python3 tools/knobbench.py --workload c4 --no-cpu-baseline --no-extras
done
