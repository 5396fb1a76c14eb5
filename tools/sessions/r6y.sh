#!/bin/bash
# round 6, GPU session Y: the default line with the cold-pair child started BEFORE the parent touches the GPU, against the stand-alone
# c3pairs command, same box, alternating
set -o pipefail
OUT=gpurun_out/r6y
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
for i in 1 2; do
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/default_$i.json" 2> "$OUT/default_$i.err"; echo "default rc=$?"
timeout -k 10 300 python3 bench.py --workload c3pairs --steps 1024 --warmup 32 --no-cpu-baseline > "$OUT/pairs_$i.json" 2> "$OUT/pairs_$i.err"; echo "pairs rc=$?"
python3 - "$OUT/default_$i.json" "$OUT/pairs_$i.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
b=d.get("batch64_hbm",{}); p=d.get("pairs64_hbm",{}); r=p.get("roofline",{})
print("default: c2k20", d["value"], "batch64", b.get("ms_per_step"), b.get("whole_step_frac_of_8TBs"), (b.get("blocks_of_256") or {}).get("whole_step_frac_of_8TBs"), "pairs64", p.get("ms_per_step"), r.get("whole_step",{}).get("frac_of_8TBs"), p.get("error"))
q=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("standalone pairs", q["ms_per_step"], q["roofline"]["whole_step"]["frac_of_8TBs"])
PY
done
