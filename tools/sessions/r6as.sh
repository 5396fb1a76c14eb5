#!/bin/bash
# round 6, GPU session AS: SQ counters of the two-level rolling pyramid pass alone (tools/pyrbench.py: 64 cold 720p frames per launch): where do its wave-cycles go?
set -o pipefail
OUT=gpurun_out/r6as
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d "$OUT/sq" -- python3 tools/pyrbench.py 12 > "$OUT/sq.stdout" 2> "$OUT/sq.stderr"; echo "rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d "$OUT/sq2" -- python3 tools/pyrbench.py 12 > "$OUT/sq2.stdout" 2> "$OUT/sq2.stderr"; echo "rc=$?"
python3 - "$OUT" <<'PY'
import csv,glob,sys,collections
for sub in ("sq","sq2"):
    fs=glob.glob(sys.argv[1]+"/"+sub+"/**/*counter_collection.csv", recursive=True)
    if not fs: print(sub,"no counter file"); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k=r["Kernel_Name"].replace("void (anonymous namespace)::","").replace("(anonymous namespace)::","").split("(")[0]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        if "pyr" in k:
            print(sub,k,{c: round(sum(x)/len(x)) for c,x in v.items()}, "dispatches", len(next(iter(v.values()))))
PY
tail -3 "$OUT/sq.stdout"
