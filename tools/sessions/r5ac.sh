#!/bin/bash
# round 5, GPU session AC: c4 -- library or harness?  round-4 harness on the new library
set -o pipefail
OUT=gpurun_out/r5ac
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
NEWLIB=$GRAFT_REPO_ROOT/accurate_aprilgroup_tracking_amd/libagt_hip.so
for i in 1 2 3; do
run c4_old_$i r04tree python3 bench.py --workload c4 --no-cpu-baseline --no-extras
run c4_new_$i . python3 bench.py --workload c4 --no-cpu-baseline --no-extras
AGT_LIB=$NEWLIB run c4_oldharness_newlib_$i r04tree python3 tools/knobbench.py --workload c4 --no-cpu-baseline --no-extras
AGT_LIB=libagt_hip.so run c4_oldharness_oldlib_$i r04tree python3 tools/knobbench.py --workload c4 --no-cpu-baseline --no-extras
done
