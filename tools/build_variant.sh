#!/bin/bash
# tools/build_variant.sh NAME [-DFLAG ...]: a reduced tuning build of the library
# (only the kernel instantiations of bench.py's default workload) as
# my-lidar-graph-slam-v2_amd/csrc/libcsm_hip_NAME.so; select it with CSM_HIP_LIB.
set -e
NAME="$1"; shift
cd "$(dirname "$0")/../my-lidar-graph-slam-v2_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wall \
    -Wno-unused-function -DCSM_FAST_BUILD "$@" -o "libcsm_hip_$NAME.so" csm_api.hip
echo "built libcsm_hip_$NAME.so"
