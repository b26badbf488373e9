#!/bin/bash
# tools/pmc_cfg5.sh TAG: FETCH_SIZE / WRITE_SIZE of configs[4]'s fine kernel (one 9.3e8-pose query), separate passes;
# prints (2 FETCH + WRITE) KB -> bytes per launch. Environment (e.g. CSM_XCD_MAP=0) is inherited.
TAG="$1"; cd "$(dirname "$0")/.."; export TMPDIR=/tmp
export CSM_BENCH_SCANS=64 CSM_BENCH_WINDOWS=64 CSM_BENCH_CONFIGS=config5
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc5_${TAG}_$C
  rocprofv3 --kernel-trace --pmc $C -d gpurun_out/pmc5_${TAG}_$C -o p --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc5_${TAG}_$C.log 2>&1 || { echo "$C failed"; tail -3 gpurun_out/pmc5_${TAG}_$C.log; exit 1; }
done
python3 - "$TAG" <<'PY'
import csv, glob, sys
tag = sys.argv[1]; tot = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = []
    for path in glob.glob("gpurun_out/pmc5_%s_%s/**/*counter_collection.csv" % (tag, C), recursive=True):
        for row in csv.DictReader(open(path)):
            if "k_score_pairs<182" in row["Kernel_Name"] and row["Counter_Name"] == C:
                vals.append(float(row["Counter_Value"]))
    tot[C] = sum(vals) / max(1, len(vals))
print(tag, tot, "HBM bytes per launch = %.3e" % ((2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024))
PY
rm -rf gpurun_out/pmc5_${TAG}_FETCH_SIZE gpurun_out/pmc5_${TAG}_WRITE_SIZE
