#!/bin/bash
# round 6, GPU session E: kernel trace of the pipelined 64-stream step (where does the step time go beside the LK launches' own duration?)
set -o pipefail
OUT=gpurun_out/r6e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d "$OUT/c3_trace" -- python3 bench.py --workload c3 --steps 64 --warmup 16 --blocks 6 --render-frames 8 --no-cpu-baseline --no-extras > "$OUT/c3_trace.stdout" 2> "$OUT/c3_trace.stderr"; echo "rc=$?"
f=$(find "$OUT/c3_trace" -name "*kernel_trace.csv" | head -1); echo "$f"; head -2 "$f"
python3 tools/trace_timeline.py "$f" > "$OUT/c3_timeline.txt"; head -60 "$OUT/c3_timeline.txt"
