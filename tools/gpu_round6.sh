#!/bin/bash
# Round-6 collection on the GPU box: the round-5 bundles with r06 file names, plus the fuzz sweep and the standing-population benches.
#   bash tools/gpu_round6.sh <tag> [bundles of tools/gpu_round5.sh ...] [fuzz] [synth]
# then, here:  bash tools/copy_profiles_r06.sh <commit> <tag>
TAG=${1:-r06}; shift || true
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
REST=()
for W in "$@"; do case "$W" in
  fuzz)  ( echo "# tools/fuzz_parity.py 48 <seed>, seeds 900-915, on the round-6 library: 720 states and 2 880 env-steps per seed across the 15 clips, HIP vs oracle"
           for sd in $(seq 900 915); do timeout 200 python tools/fuzz_parity.py 48 $sd 2>&1 | grep -v amdgpu.ids | tail -2; echo "fuzz seed $sd: rc=$?"; done ) | tee $OUT/fuzz_parity.log ;;
  synth) ( timeout 300 python tools/standing_synth_bench.py 2>&1 | grep -v amdgpu.ids | tail -6 ) | tee $OUT/standing_synth.log
         ( timeout 300 python tools/standing_step_bench.py 2>&1 | grep -v amdgpu.ids | tail -8 ) | tee $OUT/standing_step_bench.log
         ( DM_PROF_STANDING=1 DMENV_LIB=$PWD/build_ab/rprof.so timeout 300 python tools/profile_horizon.py 2>&1 | grep -v amdgpu.ids ) | tee $OUT/hstage_standing.log ;;
  *) REST+=("$W") ;;
esac; done
[ ${#REST[@]} -gt 0 ] && TAGP=r06 bash tools/gpu_round5.sh $TAG "${REST[@]}"
