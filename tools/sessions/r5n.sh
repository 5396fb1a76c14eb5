#!/bin/bash
# round 5, GPU session N: new dense clip cases (chained kernel with distinct streams); whole suite
set -o pipefail
OUT=gpurun_out/r5n
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
timeout -k 10 900 python3 -m pytest tests/test_dense.py -q -m gpu -k "rejected_and_lost" > "$OUT/dense.log" 2>&1; echo "dense rc=$?"; tail -30 "$OUT/dense.log"
timeout -k 10 1100 python3 -m pytest tests -q -m gpu -x --deselect tests/test_dense.py::test_dense_clip_with_rejected_and_lost_frames_equals_single_calls > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest.log"
