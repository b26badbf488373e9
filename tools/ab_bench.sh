#!/bin/bash
# tools/ab_bench.sh VARIANT[:ENV=VAL[,ENV=VAL]]...: the fine kernel's launch time for each tuning build /
# environment, alternating, on the box this runs on (A/B only means something on one box in one call).
cd "$(dirname "$0")/.."
export CSM_BENCH_SCANS=${CSM_BENCH_SCANS:-512}
export CSM_BENCH_WINDOWS=${CSM_BENCH_WINDOWS:-64}     # A/B numbers are per 64-window launch
for rep in 1 2; do
  for spec in "$@"; do
    v="${spec%%:*}"; envs=""
    if [ "$spec" != "$v" ]; then envs="$(echo "${spec#*:}" | tr ',' ' ')"; fi
    env $envs CSM_HIP_LIB=$PWD/my-lidar-graph-slam-v2_amd/csrc/libcsm_hip_$v.so timeout -k 10 120 python bench.py --no-configs --no-cpu-baseline --steps 5 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$spec', round(d['roofline']['avg_launch_us'],1), 'us fine;', {k: round(v,1) for k,v in d['roofline']['other_kernels_avg_us'].items()}, round(d['ms_per_step']*1e3/ (d['config']['scans_per_step']/64),1), 'us per chain;', d['config']['poses_found'])"
  done
done
