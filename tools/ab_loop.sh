#!/bin/bash
# tools/ab_loop.sh VARIANT[:ENV=VAL[,ENV=VAL]]...: the branch-and-bound batch (configs[2]) for each tuning
# build / environment, alternating, on the box this runs on.
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for spec in "$@"; do
    v="${spec%%:*}"; envs=""
    if [ "$spec" != "$v" ]; then envs="$(echo "${spec#*:}" | tr ',' ' ')"; fi
    env $envs CSM_HIP_LIB=$PWD/my-lidar-graph-slam-v2_amd/csrc/libcsm_hip_$v.so timeout -k 10 120 python bench.py --workload loop --steps 10 --no-cpu-baseline 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$spec', round(d['roofline']['avg_launch_us'],1), 'us leaf;', round(d['ms_per_step']*1e3,1), 'us per batch;', d['config']['found'])"
  done
done
