#!/bin/bash
# round 5, GPU session AH: the round-end sequence on the final tree -- build check, smoke, whole GPU suite, default bench line (wall time)
set -o pipefail
OUT=gpurun_out/r5ah
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -3 "$OUT/smoke.log"
timeout -k 10 1000 python3 -m pytest tests -q -m gpu -x > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest.log"
s=$(date +%s)
timeout -k 10 900 python3 bench.py > "$OUT/c2_default.json" 2> "$OUT/c2_default.err"; echo "bench rc=$? wall=$(( $(date +%s) - s ))s"
s=$(date +%s)
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/c2_driver.json" 2> "$OUT/c2_driver.err"; echo "bench(driver form) rc=$? wall=$(( $(date +%s) - s ))s"
python3 - "$OUT/c2_default.json" "$OUT/c2_driver.json" <<'PY'
import json,sys
for f in sys.argv[1:]:
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["roofline"]["frac"], {k:(d[k].get("value") or d[k].get("frames_per_s"),d[k].get("ms_per_step"),d[k].get("error")) for k in ("batch64_hbm","pairs64_hbm","c5_dense240")}, d["pairs64_hbm"]["roofline"]["whole_step"]["frac_of_8TBs"], d["batch64_hbm"]["whole_step_frac_of_8TBs"])
PY
