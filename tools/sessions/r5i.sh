#!/bin/bash
# round 5, GPU session I: back on the column-tile-fastest unit order (cheaper stores kept); four-wave pose role per frame vs group
set -o pipefail
OUT=gpurun_out/r5i
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
timeout -k 10 1100 python3 -m pytest tests -q -m gpu -x > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest.log"
show() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d.get("roofline",{}).get("frac"), d.get("roofline",{}).get("call_spans_us_serial_pass"))
except Exception as e: print("parse", e)
PY
}
run() { name=$1; shift; echo "== $name"; timeout -k 10 400 "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "rc=$?"; show "$OUT/$name.json"; }
P="--workload c3pairs --steps 256 --no-cpu-baseline"
run base python3 tools/knobbench.py $P
AGT_LK_LDS_PAD=10240 run lkpad10k python3 tools/knobbench.py $P
run base2 python3 tools/knobbench.py $P
AGT_LK_LDS_PAD=10240 run lkpad10k_b python3 tools/knobbench.py $P
AGT_LK_LDS_PAD=12288 run lkpad12k python3 tools/knobbench.py $P
run prod python3 bench.py $P
run c3 python3 bench.py --workload c3 --steps 256 --warmup 16 --render-frames 8 --no-cpu-baseline --no-extras
for B in 2 4 8; do for D in 4 16; do
  AGT_LIB=libagt_hip_knobs.so AGT_PNP_COOP_GROUP=0 timeout -k 10 300 python3 tools/coop240.py $B $D 2>&1 | tail -1 | sed 's/^/perframe /'
  AGT_LIB=libagt_hip_knobs.so AGT_PNP_COOP_GROUP=1 timeout -k 10 300 python3 tools/coop240.py $B $D 2>&1 | tail -1 | sed 's/^/group    /'
done; done
