#!/bin/bash
# round 6, GPU session P: (after the fill-drain ramp, stream pool) kernel trace of 256-step blocks of the pipelined 64-stream step (fill / drain against the steady state)
set -o pipefail
OUT=gpurun_out/r6p
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d "$OUT/c3_trace" -- python3 bench.py --workload c3 --steps 256 --warmup 16 --blocks 4 --render-frames 8 --no-cpu-baseline --no-extras > "$OUT/c3_trace.stdout" 2> "$OUT/c3_trace.stderr"; echo "rc=$?"
f=$(find "$OUT/c3_trace" -name "*kernel_trace.csv" | head -1)
python3 tools/trace_blocks.py "$f" > "$OUT/c3_blocks.txt"; cat "$OUT/c3_blocks.txt"; tail -c 400 "$OUT/c3_trace.stdout"
