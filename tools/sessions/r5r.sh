#!/bin/bash
# round 5, GPU session R: default bench line with the c5 extra (wall time of the whole command), twin test, whole suite
set -o pipefail
OUT=gpurun_out/r5r
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
s=$(date +%s)
timeout -k 10 900 python3 bench.py > "$OUT/c2_default.json" 2> "$OUT/c2_default.err"; echo "bench rc=$? wall=$(( $(date +%s) - s ))s"
python3 - "$OUT/c2_default.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], {k:(v.get("value"),v.get("ms_per_step"),v.get("error")) for k,v in d.get("extras",{}).items() if isinstance(v,dict)})
PY
timeout -k 10 1000 python3 -m pytest tests -q -m gpu -x > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest.log"
