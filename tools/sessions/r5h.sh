#!/bin/bash
# round 5, GPU session H: wave-major unit order + edge-free variant + cheaper stores of the two-level pass: bit-exactness, whole suite,
# cold-pair step; LK occupancy cap sweep (AGT_LK_LDS_PAD)
set -o pipefail
OUT=gpurun_out/r5h
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
for oh in 2 4 10 16; do
  AGT_TEST_LIB=libagt_hip_knobs.so AGT_PYR4=1 AGT_PYR4_OH=$oh timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "pyramid_build_all_levels or pyr_down or lk_bit_exact" > "$OUT/pyr_oh$oh.log" 2>&1; echo "pyr oh$oh rc=$?"; tail -2 "$OUT/pyr_oh$oh.log"
done
AGT_TEST_LIB=libagt_hip_knobs.so AGT_PYR4=1 AGT_PYR4_REV=0 AGT_PYR4_OH=6 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "pyramid_build_all_levels or pyr_down" > "$OUT/pyr_fwd.log" 2>&1; echo "pyr fwd rc=$?"; tail -2 "$OUT/pyr_fwd.log"
timeout -k 10 1100 python3 -m pytest tests -q -m gpu -x > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest.log"
show() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d.get("roofline",{}).get("frac"), d.get("roofline",{}).get("call_spans_us_serial_pass"))
except Exception as e: print("parse", e)
PY
}
run() { name=$1; shift; echo "== $name"; timeout -k 10 400 "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "rc=$?"; show "$OUT/$name.json"; }
P="--workload c3pairs --steps 256 --no-cpu-baseline"
run base python3 tools/knobbench.py $P
for pad in 6144 8192 10240 12288 16384 22528; do
  AGT_LK_LDS_PAD=$pad run lkpad$pad python3 tools/knobbench.py $P
done
run base2 python3 tools/knobbench.py $P
for pad in 10240 16384; do
  AGT_LK_LDS_PAD=$pad run lkpad${pad}_b python3 tools/knobbench.py $P
done
run base3 python3 tools/knobbench.py $P
run c3 python3 bench.py --workload c3 --steps 256 --warmup 16 --render-frames 8 --no-cpu-baseline --no-extras
run c2k20 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras
