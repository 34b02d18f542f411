#!/bin/bash
# usage: tools/build_ab.sh <name> [-DFLAG=VALUE ...]   -> ab_libs/<name>.so (same ABI, selected with FSPT_LIB=...)
set -e
cd "$(dirname "$0")/.."
mkdir -p ab_libs
n=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off ${AB_SLP:--fno-slp-vectorize} -fPIC -shared -std=c++17 -Wno-unused-value "$@" \
  -o ab_libs/$n.so fspt_amd/csrc/fspt_kernels.hip fspt_amd/csrc/fspt_api.cpp fspt_amd/csrc/fspt_sched_batch.cpp fspt_amd/csrc/fspt_sched_stream.cpp fspt_amd/csrc/fspt_multi.cpp fspt_amd/csrc/scene_build.cpp -ldl
echo "built ab_libs/$n.so $*"
