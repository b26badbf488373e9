#!/bin/bash
# tools/pmc_collect.sh TAG [bench args...]: SQ counters of every kernel of a short
# bench.py run, two rocprofv3 --pmc passes (8 SQ slots per pass), summarised
# into gpurun_out/TAG_sq.json. Counters and kernel trace only, as gpurun demands.
set -u
TAG="$1"; shift
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
B="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"
for P in A B; do
    eval "C=\$$P"
    rm -rf $OUT/${TAG}_pmc_$P
    rocprofv3 --kernel-trace --pmc $C -d $OUT/${TAG}_pmc_$P -o p --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs "$@" > $OUT/${TAG}_pmc_$P.log 2>&1 || { echo "pass $P failed"; tail -5 $OUT/${TAG}_pmc_$P.log; exit 1; }
done
python3 - "$TAG" <<'PY'
import csv, glob, json, os, sys
tag = sys.argv[1]
acc = {}
for P in "AB":
    for path in glob.glob("gpurun_out/%s_pmc_%s/**/*counter_collection.csv" % (tag, P), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                d = acc.setdefault(row["Kernel_Name"], {}).setdefault(row["Counter_Name"], [0.0, 0])
                d[0] += float(row["Counter_Value"]); d[1] += 1
out = {k: {c: {"mean": s / n, "dispatches": n} for c, (s, n) in cs.items()} for k, cs in acc.items()}
json.dump(out, open("gpurun_out/%s_sq.json" % tag, "w"), indent=1)
for k, cs in out.items():
    if "score" in k or "bin" in k:
        print(k[:110])
        print("   ", {c: round(v["mean"]) for c, v in sorted(cs.items())})
PY
# drop the bulky raw files, keep the summaries
rm -rf $OUT/${TAG}_pmc_A $OUT/${TAG}_pmc_B
