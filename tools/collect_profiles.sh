#!/bin/bash
# Profile collection (GPU box): kernel traces + stats of the bench commands, then one PMC counter per pass (never combined
# with other trace domains).  Usage: bash tools/collect_profiles.sh <outdir under gpurun_out>; then
# python3 tools/summarize_profiles.py gpurun_out/<outdir> r03
set -o pipefail
OUT=gpurun_out/${1:-r2prof}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
run() { name=$1; shift; echo "== $name"; timeout -k 10 400 rocprofv3 "$@" > "$OUT/$name.stdout" 2> "$OUT/$name.stderr"; echo "rc=$?"; }
run c2_stats  --kernel-trace --stats --output-format csv -d "$OUT/c2_stats"  -- python3 bench.py --no-cpu-baseline --no-extras
run c2k20_stats --kernel-trace --stats --output-format csv -d "$OUT/c2k20_stats" -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras
run c3_stats  --kernel-trace --stats --output-format csv -d "$OUT/c3_stats"  -- python3 bench.py --workload c3 --steps 64 --warmup 16 --render-frames 8 --no-cpu-baseline --no-extras
run c3pairs_stats --kernel-trace --stats --output-format csv -d "$OUT/c3pairs_stats" -- python3 bench.py --workload c3pairs --steps 100 --warmup 12 --no-cpu-baseline
run c4_stats  --kernel-trace --stats --output-format csv -d "$OUT/c4_stats"  -- python3 bench.py --workload c4 --no-cpu-baseline --no-extras
run c5_stats  --kernel-trace --stats --output-format csv -d "$OUT/c5_stats"  -- python3 bench.py --workload c5 --steps 100 --warmup 10 --no-cpu-baseline
run c2_fetch  --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/c2_fetch" -- python3 bench.py --no-cpu-baseline --no-extras --steps 192 --blocks 3
run c2_write  --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/c2_write" -- python3 bench.py --no-cpu-baseline --no-extras --steps 192 --blocks 3
run c2k20_fetch --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/c2k20_fetch" -- python3 bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 --blocks 8
run c2k20_write --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/c2k20_write" -- python3 bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 --blocks 8
run c3_fetch  --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/c3_fetch" -- python3 bench.py --workload c3 --steps 32 --warmup 16 --blocks 2 --render-frames 8 --no-cpu-baseline --no-extras
run c3_write  --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/c3_write" -- python3 bench.py --workload c3 --steps 32 --warmup 16 --blocks 2 --render-frames 8 --no-cpu-baseline --no-extras
run c3pairs_fetch --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/c3pairs_fetch" -- python3 bench.py --workload c3pairs --steps 24 --warmup 8 --blocks 2 --pair-contexts 1 --no-cpu-baseline
run c3pairs_write --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/c3pairs_write" -- python3 bench.py --workload c3pairs --steps 24 --warmup 8 --blocks 2 --pair-contexts 1 --no-cpu-baseline
run c5_fetch  --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/c5_fetch" -- python3 bench.py --workload c5 --steps 40 --warmup 10 --blocks 2 --no-cpu-baseline
run c5_write  --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/c5_write" -- python3 bench.py --workload c5 --steps 40 --warmup 10 --blocks 2 --no-cpu-baseline
run pyr_fetch --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pyr_fetch" -- python3 tools/pyrbench.py 12
run pyr_write --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pyr_write" -- python3 tools/pyrbench.py 12
ls "$OUT"
