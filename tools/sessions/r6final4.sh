#!/bin/bash
# round 6, final GPU session: the whole GPU suite once, smoke(), then the plain bench lines of every workload with cpu_baseline (profiles/r06_final_*)
set -o pipefail
OUT=gpurun_out/r6final4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1; rc=$?; tail -5 "$OUT/pytest.log"; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; rc=$?; tail -3 "$OUT/smoke.log"; echo "smoke rc=$rc"
[ $rc -ne 0 ] && exit $rc
bash tools/collect_final_benches.sh r6final4
