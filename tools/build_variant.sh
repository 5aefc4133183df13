#!/bin/bash
# build_ab/<name>.so = libdmenv.so of the working tree with extra compiler options for every translation unit; run with DMENV_LIB=build_ab/<name>.so
# (A/B inside one gpurun call):   tools/build_variant.sh keyiter '-DDM_ORDER_KEY(n,i)=((i)+((n)>>2))'     tools/build_variant.sh rprof -DDM_ROLLOUT_PROF
set -eu
cd "$(dirname "$0")/.."
mkdir -p build_ab
name=$1; shift
python deepmimic_mujoco_amd/csrc/build.py --out build_ab/$name.so "$@"
ls -la build_ab/$name.so
