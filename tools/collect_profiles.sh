#!/bin/bash
# tools/collect_profiles.sh TAG: the rocprofv3 artefacts of a round, written to gpurun_out/profiles_TAG/
# (copy what is to be judged into profiles/). Kernel trace + stats in one run; HBM counters
# (FETCH_SIZE, WRITE_SIZE) in separate --pmc runs with the kernel trace only, as gpurun demands.
set -u
TAG="$1"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/profiles_$TAG
RAW=gpurun_out/raw_$TAG
rm -rf $OUT $RAW; mkdir -p $OUT $RAW

run_stats() {   # name, bench args...
    local name="$1"; shift
    rocprofv3 --kernel-trace --stats -d $RAW/$name -o p --output-format csv -- python3 bench.py "$@" > $OUT/${name}_bench_line.json 2> $RAW/$name.err || { echo "$name failed"; tail -5 $RAW/$name.err; return 1; }
    cp "$(find $RAW/$name -name '*kernel_stats.csv' | head -1)" $OUT/${name}_kernel_stats.csv
    grep '^{' $OUT/${name}_bench_line.json | tail -1 > $OUT/${name}_bench_line.tmp && mv $OUT/${name}_bench_line.tmp $OUT/${name}_bench_line.json
    head -4 $OUT/${name}_kernel_stats.csv
}
run_pmc() {     # name, counter, bench args...
    local name="$1" counter="$2"; shift 2
    rocprofv3 --kernel-trace --pmc $counter -d $RAW/${name}_$counter -o p --output-format csv -- python3 bench.py "$@" > $RAW/${name}_$counter.log 2>&1 || { echo "$name $counter failed"; tail -5 $RAW/${name}_$counter.log; return 1; }
}

# configs[1], the default line
run_stats csm --no-configs --no-cpu-baseline || exit 1
export CSM_BENCH_SCANS=512 CSM_BENCH_DISTINCT=256
run_pmc csm FETCH_SIZE --steps 3 --warmup 1 --no-configs --no-cpu-baseline || exit 1
run_pmc csm WRITE_SIZE --steps 3 --warmup 1 --no-configs --no-cpu-baseline || exit 1
# configs[4]: the one workload whose maps leave L2 (56 MB pair-row copy, 8 MB grid)
run_stats cfg5 --workload config5 --steps 5 --warmup 1 || exit 1
run_pmc cfg5 FETCH_SIZE --workload config5 --steps 2 --warmup 1 || exit 1
run_pmc cfg5 WRITE_SIZE --workload config5 --steps 2 --warmup 1 || exit 1
unset CSM_BENCH_SCANS CSM_BENCH_DISTINCT CSM_BENCH_CONFIGS CSM_BENCH_WINDOWS
# configs[2]: the branch-and-bound batch
run_stats loop --workload loop --steps 10 --no-cpu-baseline || exit 1
run_pmc loop FETCH_SIZE --workload loop --steps 3 --warmup 1 --no-cpu-baseline || exit 1
run_pmc loop WRITE_SIZE --workload loop --steps 3 --warmup 1 --no-cpu-baseline || exit 1

python3 - "$TAG" <<'PY'
import csv, glob, json, os, sys
tag = sys.argv[1]
out, raw = "gpurun_out/profiles_%s" % tag, "gpurun_out/raw_%s" % tag
sys.path.insert(0, "my-lidar-graph-slam-v2_amd")
from csm_hip import _lib
version = _lib.load().csm_version().decode()
def counters(name):
    acc = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        for path in glob.glob("%s/%s_%s/**/*counter_collection.csv" % (raw, name, counter), recursive=True):
            for row in csv.DictReader(open(path)):
                d = acc.setdefault(row["Kernel_Name"], {}).setdefault(row["Counter_Name"], [0.0, 0])
                d[0] += float(row["Counter_Value"]); d[1] += 1
    return {k: {c: {"mean_KB": s / n, "dispatches": n} for c, (s, n) in v.items()} for k, v in acc.items()}
for name, pick, wpl in (("csm", "_batch<", int(os.environ.get("CSM_BENCH_WINDOWS", "256"))), ("cfg5", "k_score_pairs<", 1),
                        ("loop", "_batch<", 256)):
    per = counters(name)
    if name == "cfg5":      # the coarse-first search: the pair kernel on the phase-major copy + the fine work list
        dom = [k for k in per if "k_score_pairs<" in k or "k_score_pairs_list<" in k or "k_score_joint_one<" in k]
    else:                   # the joint kernels of a batch: fp32 bound pass (if any) + exact kernel
        dom = [k for k in per if "k_score_joint" in k]
    doc = {"library_version": version, "windows_per_launch": wpl, "per_kernel": per,
           "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over a short "
                     "bench.py run; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: FETCH_SIZE doubled per the gfx950 note "
                     "in MI355X_MICROARCH.md (HBM section)"}
    if dom:
        # the level is several launches (bound pass and exact kernel, each once per row-block shape;
        # coarse pass + fine work list for the coarse-first search): its bytes are their sum
        dom = sorted(dom)
        f = sum(per[k].get("FETCH_SIZE", {}).get("mean_KB", 0.0) for k in dom)
        w = sum(per[k].get("WRITE_SIZE", {}).get("mean_KB", 0.0) for k in dom)
        doc.update(kernel=" + ".join(dom), FETCH_SIZE_KB=f, WRITE_SIZE_KB=w, hbm_bytes_per_launch=(2 * f + w) * 1024)
    json.dump(doc, open("%s/%s_pmc_traffic.json" % (out, name), "w"), indent=1)
    print(name, doc.get("kernel"), doc.get("hbm_bytes_per_launch"))
PY
rm -rf $RAW
ls -la $OUT
