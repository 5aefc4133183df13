#!/bin/bash
# A/B of two builds of libdmenv.so inside ONE gpurun call (same box, alternating runs): how every kernel change of round 2 was judged.
#   here (no GPU):   tools/ab_bench.sh build <git-ref-A> [<git-ref-B, default: the working tree>]     -> build_ab/A.so, build_ab/B.so
#   on the GPU box:  gpurun -- 'bash tools/ab_bench.sh run [bench.py arguments]'                      -> alternating A / B bench lines
# build_ab/ is git-ignored but travels to the GPU box with the snapshot.  Both builds must export the C ABI the working tree's _abi.py binds
# (a ref from before an entry point was added fails to load: its line then shows the loader's message).
set -eu
cd "$(dirname "$0")/.."
LIB=deepmimic_mujoco_amd/csrc/libdmenv.so
case "${1:-}" in
  build)
    mkdir -p build_ab
    A=${2:?git ref of build A}; B=${3:-}
    cp $LIB build_ab/keep.so 2>/dev/null || true
    build_ref() {   # $1 ref ('' = working tree), $2 output
      if [ -n "$1" ]; then
        tmp=$(mktemp -d); git archive "$1" deepmimic_mujoco_amd/csrc include | tar -x -C "$tmp"
        if [ -f "$tmp/deepmimic_mujoco_amd/csrc/kernels_packed.hip" ]; then python "$tmp/deepmimic_mujoco_amd/csrc/build.py" --out "$PWD/$2"
        else /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-value -Wno-shift-count-negative \
          -Wno-implicit-const-int-float-conversion -I"$tmp/include" -I"$tmp/deepmimic_mujoco_amd/csrc" "$tmp/deepmimic_mujoco_amd/csrc/dmenv.hip" -o "$2"; fi   # (refs from before the two-unit build)
        rm -rf "$tmp"
      else
        python deepmimic_mujoco_amd/csrc/build.py --out "$PWD/$2"
      fi
    }
    build_ref "$A" build_ab/A.so & build_ref "$B" build_ab/B.so & wait
    ls -la build_ab/A.so build_ab/B.so ;;
  run)
    shift
    cp $LIB build_ab/keep.so
    for rep in 1 2; do for v in A B; do
      cp build_ab/$v.so $LIB
      out=$(timeout 300 python bench.py --no-pmc --no-cpu-baseline --no-gym-loop "$@" 2>build_ab/err.txt | cut -c1-110)
      echo "$v ${out:-$(tail -1 build_ab/err.txt | cut -c1-160)}"
    done; done
    cp build_ab/keep.so $LIB ;;
  *) sed -n 2,6p "$0" ;;
esac
