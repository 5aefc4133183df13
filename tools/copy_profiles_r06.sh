#!/bin/bash
# gpurun_out/<tag>/ (tools/gpu_round5.sh bundles, TAGP=r06) -> profiles/r06_* :  bash tools/copy_profiles_r06.sh <commit the call ran on> <tag> [<tag2> ...]
# Every markdown summary gets a first line naming the commit its numbers were measured on; the JSON lines get a sidecar entry in profiles/r06_commits.txt.
set -u
P=profiles; C=$1; shift
stamp() { ( echo "<!-- measured on commit $C (gpurun call $2) -->"; grep -v amdgpu.ids "$1" ) > "$3"; }
for T in "$@"; do O=gpurun_out/$T
  for f in cfg3 cfg3_alive cfg3_v3-config cfg3_driver_window cfg3_driver_command cfg3_unqueued cfg2 rollout_fused standing; do
    [ -f $O/bench_$f.json ] && cp $O/bench_$f.json $P/r06_bench_$f.json && echo "r06_bench_$f.json $C $T" >> $P/r06_commits.txt; done
  [ -f $O/bench_cfg4.json ] && cp $O/bench_cfg4.json $P/r06_bench_cfg4_shard.json && echo "r06_bench_cfg4_shard.json $C $T" >> $P/r06_commits.txt
  [ -f $O/bench_cfg5.json ] && cp $O/bench_cfg5.json $P/r06_bench_cfg5_shard.json && echo "r06_bench_cfg5_shard.json $C $T" >> $P/r06_commits.txt
  [ -f $O/pytest_gpu.log ] && ( echo "<!-- measured on commit $C (gpurun call $T) -->"; grep -v amdgpu.ids $O/pytest_gpu.log; cat $O/smoke.log ) > $P/r06_gpu_tests.md
  [ -f $O/krollout_summary.md ] && stamp $O/krollout_summary.md $T $P/r06_krollout_summary.md
  [ -f $O/kstep_summary.md ] && stamp $O/kstep_summary.md $T $P/r06_kstep_summary.md
  [ -f $O/hstage.log ] && stamp $O/hstage.log $T $P/r06_horizon_stage_cycles.md
  [ -f $O/stage.log ] && stamp $O/stage.log $T $P/r06_packed_stage_cycles.md
  [ -f $O/fuzz_parity.log ] && stamp $O/fuzz_parity.log $T $P/r06_fuzz_parity.log
  [ -f $O/standing_synth.log ] && stamp $O/standing_synth.log $T $P/r06_standing_synth.log
  [ -f $O/standing_step_bench.log ] && stamp $O/standing_step_bench.log $T $P/r06_standing_step_bench.log
  [ -f $O/hstage_standing.log ] && stamp $O/hstage_standing.log $T $P/r06_horizon_stage_cycles_standing.md
  [ -f $O/trpo_train_60s.json ] && cp $O/trpo_train_60s.json $P/r06_trpo_learning_curve.json && echo "r06_trpo_learning_curve.json $C $T" >> $P/r06_commits.txt
  [ -f $O/trpo_update_profile.json ] && cp $O/trpo_update_profile.json $P/r06_trpo_update_profile_native.json && echo "r06_trpo_update_profile_native.json $C $T" >> $P/r06_commits.txt
  [ -f $O/trpo_imitation_60s.json ] && cp $O/trpo_imitation_60s.json $P/r06_trpo_imitation_curve.json && echo "r06_trpo_imitation_curve.json $C $T" >> $P/r06_commits.txt
  if [ -d $O/raw ]; then cp $O/raw/r06_* $P/raw/ 2>/dev/null; fi
done
tac $P/r06_commits.txt | awk '!s[$1]++' | tac > $P/.r06c && mv $P/.r06c $P/r06_commits.txt
ls $P | grep r06
