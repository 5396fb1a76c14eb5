#!/bin/bash
# round 5, GPU session F: the shipped two-level rolling pass + device guard: whole GPU suite, then every workload's bench line
set -o pipefail
OUT=gpurun_out/r5f
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
timeout -k 10 1100 python3 -m pytest tests -q -m gpu -x > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -5 "$OUT/pytest.log"
show() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d.get("roofline",{}).get("frac"), d.get("roofline",{}).get("whole_step"))
except Exception as e: print("parse", e)
PY
}
run() { name=$1; shift; echo "== $name"; timeout -k 10 400 "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "rc=$?"; show "$OUT/$name.json"; }
run pairs python3 bench.py --workload c3pairs --steps 256 --no-cpu-baseline
run pairs_ctx5 python3 bench.py --workload c3pairs --steps 256 --no-cpu-baseline --pair-contexts 5
run pairs_ctx6 python3 bench.py --workload c3pairs --steps 256 --no-cpu-baseline --pair-contexts 6
run c2 python3 bench.py --no-cpu-baseline --no-extras
run c2k20 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras
AGT_PYR4=0 run c2_nop4 python3 tools/knobbench.py --no-cpu-baseline --no-extras
AGT_PYR4=0 run c2k20_nop4 python3 tools/knobbench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras
run c2_knobs python3 tools/knobbench.py --no-cpu-baseline --no-extras
run c2k20_knobs python3 tools/knobbench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras
run c3 python3 bench.py --workload c3 --steps 256 --warmup 16 --render-frames 8 --no-cpu-baseline --no-extras
AGT_PYR4=0 run c3_nop4 python3 tools/knobbench.py --workload c3 --steps 256 --warmup 16 --render-frames 8 --no-cpu-baseline --no-extras
run c3_knobs python3 tools/knobbench.py --workload c3 --steps 256 --warmup 16 --render-frames 8 --no-cpu-baseline --no-extras
run c4 python3 bench.py --workload c4 --no-cpu-baseline --no-extras
run c5 python3 bench.py --workload c5 --no-cpu-baseline
run pairs2 python3 bench.py --workload c3pairs --steps 256 --no-cpu-baseline
