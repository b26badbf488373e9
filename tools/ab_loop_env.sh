#!/bin/bash
# tools/ab_loop_env.sh "ENV=VAL ..." ...: the branch-and-bound batch (configs[2]; CSM_LOOP_QUERIES=2048 for
# configs[3] on one GPU) of the full library under each environment, alternating; "-" = none.
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for spec in "$@"; do
    envs=""; [ "$spec" != "-" ] && envs="$spec"
    env $envs timeout -k 10 200 python bench.py --workload ${CSM_AB_WORKLOAD:-loop} --steps ${CSM_AB_STEPS:-10} --no-cpu-baseline 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$spec]', round(d['roofline']['avg_launch_us'],1), 'us leaf;', round(d['ms_per_step']*1e3,1), 'us per batch;', d['config'].get('found'))"
  done
done
