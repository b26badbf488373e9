#!/bin/bash
# tools/ab_env.sh "ENV=VAL ENV=VAL" ...: the default bench line (full library) under each environment,
# alternating, two rounds; prints the fine kernel's launch time. "-" = no extra environment.
cd "$(dirname "$0")/.."
export CSM_BENCH_SCANS=${CSM_BENCH_SCANS:-512}
export CSM_BENCH_WINDOWS=${CSM_BENCH_WINDOWS:-64}
for rep in 1 2; do
  for spec in "$@"; do
    envs=""; [ "$spec" != "-" ] && envs="$spec"
    env $envs timeout -k 10 120 python bench.py --no-configs --no-cpu-baseline --steps 5 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$spec]', round(d['roofline']['avg_launch_us'],1), 'us fine;', d['roofline']['kernel'][:40], {k: round(v,1) for k,v in d['roofline']['other_kernels_avg_us'].items()}, d['config']['poses_found'])"
  done
done
