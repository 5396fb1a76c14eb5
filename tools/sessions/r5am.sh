#!/bin/bash
# round 5, GPU session AM: synchronous host-array entry points (agt_solve_pnp_host / agt_project_points_host): whole suite + per-call latencies
set -o pipefail
OUT=gpurun_out/r5am
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
timeout -k 10 1000 python3 -m pytest tests -q -m gpu -x > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -6 "$OUT/pytest.log"
timeout -k 10 600 python3 bench.py --no-cpu-baseline > "$OUT/c2.json" 2> "$OUT/c2.err"; echo "bench rc=$?"
python3 - "$OUT/c2.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["per_call_latency_us"], d.get("drop_in_frame_us_median"), d.get("live_frame_us_median"))
PY
