#!/bin/bash
# round 6, GPU session AB: the split pipeline's own residency cap (10 per CU) at other stream counts: 32 / 48 / 96 / 128 streams, library's choice
# against no cap, same box
set -o pipefail
OUT=gpurun_out/r6ab
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
show() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["timing"]["ms_per_step_p10"], d.get("accepted_frac"))
except Exception as e: print("parse", e)
PY
}
run() { name=$1; shift; echo "== $name"; timeout -k 10 300 python3 bench.py --no-cpu-baseline --workload c3 --steps 256 --render-frames 8 "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "rc=$?"; show "$OUT/$name.json"; }
for B in 32 48 96 128; do
for i in 1 2; do
run b${B}_auto_$i --streams $B
run b${B}_none_$i --streams $B --stream-lk-cu 0
done
done
