#!/bin/bash
# tools/ab_cfg5.sh ENV=VAL ...: configs[4] (one 9.3e8-pose query) once per environment setting, twice round
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for spec in "$@"; do
    envs=""; [ "$spec" != "default" ] && envs="$(echo "$spec" | tr ',' ' ')"
    env $envs CSM_BENCH_SCANS=64 CSM_BENCH_WINDOWS=64 CSM_BENCH_CONFIGS=config5 timeout -k 10 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['configs']['config5']; print('$spec', round(d['roofline']['avg_launch_us']/1e3,2), 'ms kernel;', round(d['ms_per_query'],2), 'ms per query; found', d['found'])"
  done
done
