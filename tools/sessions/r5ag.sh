#!/bin/bash
# round 5, GPU session AG: does the chip hold its engine clock under the 64-stream / 64-pair steps?  (rocm-smi polled beside long runs)
set -o pipefail
OUT=gpurun_out/r5ag
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
rocm-smi -d 0 --showclocks --showpower --showuse > "$OUT/smi_idle.txt" 2>&1; echo "smi rc=$?"; head -30 "$OUT/smi_idle.txt"
timeout -k 10 200 python3 tools/clockwatch.py -- python3 bench.py --workload c3 --steps 4096 --warmup 16 --blocks 12 --render-frames 8 --no-cpu-baseline > "$OUT/c3.json" 2> "$OUT/c3.err"; echo "rc=$?"; cat "$OUT/c3.json"
timeout -k 10 200 python3 tools/clockwatch.py -- python3 bench.py --workload c3pairs --steps 4096 --blocks 12 --no-cpu-baseline > "$OUT/c3pairs.json" 2> "$OUT/c3pairs.err"; echo "rc=$?"; cat "$OUT/c3pairs.json"
timeout -k 10 200 python3 tools/clockwatch.py -- python3 bench.py --steps 8000 --blocks 12 --no-cpu-baseline --no-extras > "$OUT/c2.json" 2> "$OUT/c2.err"; echo "rc=$?"; cat "$OUT/c2.json"
