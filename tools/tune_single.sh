#!/bin/bash
# tools/tune_single.sh: one-query-per-call latency of configs[1] for tile-split factors and block shapes
cd "$(dirname "$0")/.."
for rep in 1 2; do
for spec in "default" "CSM_FINE_SLICES=1" "CSM_FINE_SLICES=2" "CSM_FINE_SLICES=3" "CSM_FINE_SLICES=4" "CSM_FINE_SLICES=8" \
            "CSM_FINE_SLICES=1,CSM_PAIR_GROUPS=3" "CSM_FINE_SLICES=1,CSM_PAIR_GROUPS=2" "CSM_FINE_SLICES=2,CSM_PAIR_GROUPS=3" \
            "CSM_FINE_SLICES=1,CSM_PAIR_R=6,CSM_PAIR_GROUPS=4" "CSM_FINE_SLICES=1,CSM_PAIR_NCBX=2,CSM_PAIR_GROUPS=6"; do
  envs=""; [ "$spec" != "default" ] && envs="$(echo "$spec" | tr ',' ' ')"
  echo -n "$spec: "; env $envs timeout -k 10 100 python tools/t_single.py 2>&1 | grep median
done
done
