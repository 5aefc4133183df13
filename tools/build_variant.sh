#!/bin/bash
# build_ab/<name>.so = libdmenv.so of the working tree with extra defines; run with DMENV_LIB=build_ab/<name>.so (A/B inside one gpurun call)
#   tools/build_variant.sh keyiter '-DDM_ORDER_KEY(n,i)=((i)+((n)>>2))'
set -eu
cd "$(dirname "$0")/.."
# the product build's backend options (csrc/build.py BACKEND_FLAGS); DM_BACKEND_FLAGS= (empty) builds without them
DM_BACKEND_FLAGS=${DM_BACKEND_FLAGS--mllvm -amdgpu-sched-strategy=max-ilp -mllvm -greedy-regclass-priority-trumps-globalness=1}
mkdir -p build_ab
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-value -Wno-shift-count-negative -Wno-implicit-const-int-float-conversion $DM_BACKEND_FLAGS \
  "$@" -Iinclude -Ideepmimic_mujoco_amd/csrc deepmimic_mujoco_amd/csrc/dmenv.hip -o build_ab/$name.so
ls -la build_ab/$name.so
