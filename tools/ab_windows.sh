#!/bin/bash
# tools/ab_windows.sh N ...: configs[1] with N windows per launch chain
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for w in "$@"; do
    CSM_BENCH_WINDOWS=$w CSM_BENCH_SCANS=1024 CSM_BENCH_DISTINCT=512 timeout -k 10 200 python bench.py --no-configs --no-cpu-baseline --steps 4 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w windows per chain:', round(d['value']/1e10,4), 'e10 poses/s;', round(d['roofline']['avg_launch_us'],1), 'us fine kernel;', d['config']['poses_found'])"
  done
done
