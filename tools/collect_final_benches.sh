#!/bin/bash
# Plain (un-profiled) bench lines of every workload, with cpu_baseline: bash tools/collect_final_benches.sh <outdir under gpurun_out>
OUT=gpurun_out/${1:-final}
mkdir -p "$OUT"
run() { name=$1; shift; echo "== $name"; timeout -k 10 600 python3 bench.py "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "rc=$?"; }
run c2
run c2k20 --steps 20 --warmup 5
run c3 --workload c3 --steps 600 --warmup 16 --render-frames 8
run c3_blocks256 --workload c3 --steps 256 --warmup 16 --render-frames 8 --no-cpu-baseline
run c3pairs --workload c3pairs --steps 1024 --warmup 32
run c4 --workload c4 --no-extras
run c5 --workload c5
