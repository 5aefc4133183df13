#!/bin/bash
# gpurun_out/<tag>/ (tools/gpu_round4.sh bundles tests bench trace benchall standing train) -> profiles/r04_* :  bash tools/copy_profiles_r04.sh <tag> [<tag2> ...]
set -u
P=profiles
for T in "$@"; do O=gpurun_out/$T
  for f in cfg3 cfg3_alive cfg3_v3-config cfg3_driver_window cfg3_unqueued cfg2 rollout_fused standing; do [ -f $O/bench_$f.json ] && cp $O/bench_$f.json $P/r04_bench_$f.json; done
  [ -f $O/bench_cfg4.json ] && cp $O/bench_cfg4.json $P/r04_bench_cfg4_shard.json; [ -f $O/bench_cfg5.json ] && cp $O/bench_cfg5.json $P/r04_bench_cfg5_shard.json
  [ -f $O/pytest_gpu.log ] && ( grep -v amdgpu.ids $O/pytest_gpu.log; cat $O/smoke.log ) > $P/r04_gpu_tests.md
  [ -f $O/krollout_summary.md ] && cp $O/krollout_summary.md $P/r04_krollout_summary.md; [ -f $O/kstep_summary.md ] && cp $O/kstep_summary.md $P/r04_kstep_summary.md
  [ -f $O/trpo_train_60s.json ] && cp $O/trpo_train_60s.json $P/r04_trpo_learning_curve.json; [ -f $O/trpo_update_profile.json ] && cp $O/trpo_update_profile.json $P/r04_trpo_update_profile_native.json
  [ -f $O/trpo_imitation_60s.json ] && cp $O/trpo_imitation_60s.json $P/r04_trpo_imitation_curve.json; [ -f $O/train_kernels.md ] && cp $O/train_kernels.md $P/r04_train_kernels.md
  [ -f $O/stage.log ] && grep -v amdgpu.ids $O/stage.log > $P/r04_packed_stage_cycles.md; [ -f $O/hstage.log ] && grep -v amdgpu.ids $O/hstage.log > $P/r04_horizon_stage_cycles.md
  if [ -d $O/raw ]; then for f in $O/raw/*; do cp $f $P/raw/r04_cfg3_queue_$(basename $f); done; fi
done
python tools/kernel_resources.py deepmimic_mujoco_amd/csrc/libdmenv.so $P/r04_kernel_resources.md > /dev/null 2>&1
ls $P | grep r04
