#!/bin/bash
# round 6, GPU session S: extra LDS per one-wave LK workgroup (knobs build, AGT_LK_LDS_PAD) = resident LK waves per CU, c3 (base 7,680 B with
# the 5-px margin: 163,840 / (7,680 + pad) workgroups per CU if LDS is what limits them)
set -o pipefail
OUT=gpurun_out/r6s
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
show() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]; print(d["ms_per_step"], d["timing"]["ms_per_step_p10"], r["whole_step"]["frac_of_8TBs"], r["avg_launch_us"], d["config"].get("separate_kernel_spans_us") or r.get("separate_kernel_spans_us"))
except Exception as e: print("parse", e)
PY
}
run() { name=$1; shift; echo "== $name"; timeout -k 10 300 python3 tools/knobbench.py --no-cpu-baseline --workload c3 --steps 256 "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "rc=$?"; show "$OUT/$name.json"; }
for pad in 0 2560 4096 5120 6144 7168 8192 9216 10240 12288 0; do
AGT_LK_LDS_PAD=$pad run pad_$pad
done
