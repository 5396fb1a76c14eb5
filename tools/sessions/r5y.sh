#!/bin/bash
# round 5, GPU session Y: tilted-sensor model -- its tests, the whole suite, and every workload's line against the round-4 tree (the
# distortion branch gained a pointer test per projection; the benches' cameras have no distortion)
set -o pipefail
OUT=gpurun_out/r5y
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
timeout -k 10 600 python3 -m pytest tests/test_gpu_tilt.py tests/test_gpu_parity.py -q -m gpu -x > "$OUT/pytest_tilt.log" 2>&1; echo "tilt rc=$?"; tail -15 "$OUT/pytest_tilt.log"
timeout -k 10 1000 python3 -m pytest tests -q -m gpu -x > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest.log"
show() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])
except Exception as e: print("parse", e)
PY
}
run() { name=$1; dir=$2; shift 2; echo "== $name"; (cd $dir && timeout -k 10 400 "$@") > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "rc=$?"; show "$OUT/$name.json"; }
for i in 1 2; do
run c2k20_new_$i . python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras
run c2k20_old_$i r04tree python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras
run c2_new_$i . python3 bench.py --no-cpu-baseline --no-extras
run c5_new_$i . python3 bench.py --workload c5 --no-cpu-baseline
run c5_old_$i r04tree python3 bench.py --workload c5 --no-cpu-baseline
run c4_new_$i . python3 bench.py --workload c4 --no-cpu-baseline --no-extras
run c3_new_$i . python3 bench.py --workload c3 --steps 256 --warmup 16 --render-frames 8 --no-cpu-baseline
run c3pairs_new_$i . python3 bench.py --workload c3pairs --steps 256 --no-cpu-baseline
done
for B in 2 8; do
echo "== coop240 B=$B"; AGT_LIB=libagt_hip.so timeout -k 10 300 python3 tools/coop240.py $B 16 > "$OUT/coop240_B$B.txt" 2>&1; echo "rc=$?"; tail -1 "$OUT/coop240_B$B.txt"
done
