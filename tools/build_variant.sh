#!/bin/bash
# tools/build_variant.sh NAME [-DFLAG ...]: a reduced tuning build of the library
# (only the kernel instantiations of bench.py's default workload) as
# my-lidar-graph-slam-v2_amd/csrc/libcsm_hip_NAME.so; select it with CSM_HIP_LIB.
set -e
NAME="$1"; shift
cd "$(dirname "$0")/../my-lidar-graph-slam-v2_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function -DCSM_FAST_BUILD -DCSM_TUNING"
OBJS=""
for u in csm_api csm_plan csm_window csm_batch csm_launch csm_joint_kernels csm_phase_kernels csm_map_api csm_cost_api csm_group; do
    /opt/rocm/bin/hipcc $FLAGS "$@" -c -o "v_$NAME.$u.o" $u.hip &
    OBJS="$OBJS v_$NAME.$u.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "libcsm_hip_$NAME.so" $OBJS
rm -f $OBJS
echo "built libcsm_hip_$NAME.so"
