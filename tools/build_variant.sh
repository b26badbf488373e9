#!/bin/bash
# tools/build_variant.sh NAME [-DFLAG ...]: a reduced tuning build of the library
# (only the kernel instantiations of bench.py's default workload) as
# my-lidar-graph-slam-v2_amd/csrc/libcsm_hip_NAME.so; select it with CSM_HIP_LIB.
set -e
NAME="$1"; shift
cd "$(dirname "$0")/../my-lidar-graph-slam-v2_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function -DCSM_FAST_BUILD -DCSM_TUNING"
/opt/rocm/bin/hipcc $FLAGS "$@" -c -o "v_$NAME.api.o" csm_api.hip &
/opt/rocm/bin/hipcc $FLAGS "$@" -c -o "v_$NAME.joint.o" csm_joint_kernels.hip &
/opt/rocm/bin/hipcc $FLAGS "$@" -c -o "v_$NAME.phase.o" csm_phase_kernels.hip &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "libcsm_hip_$NAME.so" "v_$NAME.api.o" "v_$NAME.joint.o" "v_$NAME.phase.o"
rm -f "v_$NAME.api.o" "v_$NAME.joint.o" "v_$NAME.phase.o"
echo "built libcsm_hip_$NAME.so"
