#!/bin/bash
# round 6, GPU session K: c4 (one 1920x1080 stream) on the round-4 tree (git archive b599ebb, built in place under r04tree/)
# against today's tree, same box: the bench lines (three alternating runs) and a kernel trace of each (which kernel, or the host?)
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6k
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
show() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["timing"]["ms_per_step_p10"], d["timing"]["ms_per_step_p90"])
except Exception as e: print("parse", e)
PY
}
for i in 1 2 3; do
  for t in r04 now; do
    d=$GRAFT_REPO_ROOT; [ $t = r04 ] && d=$GRAFT_REPO_ROOT/r04tree
    echo "== c4_${t}_$i"; (cd $d && timeout -k 10 300 python3 bench.py --workload c4 --no-cpu-baseline --no-extras > "$OUT/c4_${t}_$i.json" 2> "$OUT/c4_${t}_$i.err"); echo "rc=$?"; show "$OUT/c4_${t}_$i.json"
  done
done
for t in r04 now; do
  d=$GRAFT_REPO_ROOT; [ $t = r04 ] && d=$GRAFT_REPO_ROOT/r04tree
  (cd $d && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$t" -- python3 bench.py --workload c4 --no-cpu-baseline --no-extras --blocks 6 > "$OUT/trace_$t.stdout" 2> "$OUT/trace_$t.stderr"); echo "trace $t rc=$?"
  f=$(find "$OUT/trace_$t" -name "*kernel_stats.csv" | head -1); head -6 "$f"
done
