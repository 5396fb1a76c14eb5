#!/bin/bash
# round 6, GPU session W: residency cap (ABI 503), pair build, refresh-interval blocks: full suite, default bench line
set -o pipefail
OUT=gpurun_out/r6w
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1; rc=$?; tail -5 "$OUT/pytest.log"; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc=$?"
python3 - "$OUT/bench_default.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("c2k20", d["value"], d["ms_per_step"], d["roofline"]["frac"])
b=d.get("batch64_hbm",{}); print("batch64", b.get("ms_per_step"), b.get("whole_step_frac_of_8TBs"), b.get("steps_per_block"), b.get("blocks_of_256"), b.get("accepted_frac"))
p=d.get("pairs64_hbm",{}); r=p.get("roofline",{}); print("pairs64", p.get("ms_per_step"), r.get("frac"), r.get("avg_launch_us"), r.get("alone"), r.get("whole_step",{}).get("frac_of_8TBs"))
c=d.get("c5_dense240",{}); print("c5", c.get("value"))
print(json.dumps(d.get("per_call_latency_us",{}).get("c_abi_call_only")))
PY
