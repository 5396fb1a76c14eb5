#!/bin/bash
# The c2 passes of tools/collect_profiles.sh only (after a change of the fused step): stats of the default and of the driver-style
# invocation, FETCH_SIZE / WRITE_SIZE at 32 and 20 frames per launch.  Usage: bash tools/collect_profiles_c2.sh <outdir under gpurun_out>
set -o pipefail
OUT=gpurun_out/${1:-r2prof}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
run() { name=$1; shift; echo "== $name"; timeout -k 10 400 rocprofv3 "$@" > "$OUT/$name.stdout" 2> "$OUT/$name.stderr"; echo "rc=$?"; }
run c2_stats  --kernel-trace --stats --output-format csv -d "$OUT/c2_stats"  -- python3 bench.py --no-cpu-baseline
run c2k20_stats --kernel-trace --stats --output-format csv -d "$OUT/c2k20_stats" -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras
run c2_fetch  --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/c2_fetch" -- python3 bench.py --no-cpu-baseline --no-extras --steps 192 --blocks 3
run c2_write  --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/c2_write" -- python3 bench.py --no-cpu-baseline --no-extras --steps 192 --blocks 3
run c2d20_fetch --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/c2d20_fetch" -- python3 bench.py --no-cpu-baseline --no-extras --steps 200 --depth 20 --blocks 3
run c2d20_write --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/c2d20_write" -- python3 bench.py --no-cpu-baseline --no-extras --steps 200 --depth 20 --blocks 3
ls "$OUT"
