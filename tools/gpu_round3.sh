#!/bin/bash
# Round-3 gpurun bundle (run from the repo root on the GPU box):  bash tools/gpu_round3.sh <tag> [tests|bench|all]
# Outputs under gpurun_out/<tag>/ ; raw rocprofv3 exports (CSV, trimmed) under gpurun_out/<tag>/raw/ -> copy into profiles/raw/.
set -u
TAG=${1:-r03}; WHAT=${2:-all}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT/raw
export TMPDIR=/tmp
if [ "$WHAT" = tests ] || [ "$WHAT" = all ]; then
  ( timeout 2400 python -m pytest tests -m gpu -x -q --durations=12 2>&1 | tail -60 ) > $OUT/pytest_gpu.log
  tail -25 $OUT/pytest_gpu.log
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.log
fi
if [ "$WHAT" = bench ] || [ "$WHAT" = all ]; then
  DM_PROFILE_KEEP=$OUT/raw timeout 900 python bench.py > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err; cut -c1-400 $OUT/bench_cfg3.json
  for rw in alive v3-config; do
    timeout 300 python bench.py --reward $rw --no-pmc --no-cpu-baseline --no-gym-loop > $OUT/bench_cfg3_$rw.json 2> $OUT/bench_cfg3_$rw.err; cut -c1-200 $OUT/bench_cfg3_$rw.json
  done
  for wl in cfg4 cfg2; do
    timeout 300 python bench.py --workload $wl --no-pmc --no-cpu-baseline --no-gym-loop > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err; cut -c1-200 $OUT/bench_$wl.json
  done
  # configs[4] shard (8192 envs): DPVecEnv picks four environments per wavefront; with the live PMC passes of that kernel; and pinned to the one-env kernel
  mkdir -p $OUT/raw_packed
  DM_PROFILE_KEEP=$OUT/raw_packed timeout 600 python bench.py --workload cfg5 --no-cpu-baseline --no-gym-loop > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err; cut -c1-200 $OUT/bench_cfg5.json
  timeout 300 python bench.py --workload cfg5 --packed 0 --no-pmc --no-cpu-baseline --no-gym-loop > $OUT/bench_cfg5_one_env_per_wave.json 2>/dev/null; cut -c1-160 $OUT/bench_cfg5_one_env_per_wave.json
  for n in 4096 16384 32768; do timeout 300 python bench.py --envs $n --packed 1 --pipeline $([ $n = 4096 ] && echo 1 || echo 2) --no-pmc --no-cpu-baseline --no-gym-loop > $OUT/bench_cfg3_packed_$n.json 2>/dev/null; cut -c1-160 $OUT/bench_cfg3_packed_$n.json; done
  timeout 300 python bench.py --envs 16384 --packed 0 --no-pmc --no-cpu-baseline --no-gym-loop > $OUT/bench_cfg3_one_env_16384.json 2>/dev/null; cut -c1-160 $OUT/bench_cfg3_one_env_16384.json
  timeout 300 python bench.py --workload cfg5 --no-reorder --no-pmc --no-cpu-baseline --no-gym-loop > $OUT/bench_cfg5_no_reorder.json 2>/dev/null; cut -c1-120 $OUT/bench_cfg5_no_reorder.json
  timeout 300 python tools/profile_packed.py > $OUT/packed_stage_cycles.txt 2>&1; head -30 $OUT/packed_stage_cycles.txt
  timeout 300 python tools/profile_stages.py > $OUT/stage_cycles.txt 2>&1; head -12 $OUT/stage_cycles.txt
  [ -x tools/ubench/lone ] && tools/ubench/lone > $OUT/ubench_lone.txt 2>&1; cat $OUT/ubench_lone.txt
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/p_trace5 -- python $OLDPWD/bench.py --workload cfg5 --steps 96 --warmup 16 --no-pmc --no-cpu-baseline --no-gym-loop > /dev/null 2>&1 )
  ROWS=6 python tools/rocprof_summary.py $OUT/kstep_packed_summary.md "step kernels — $TAG, MI355X (bench.py --workload cfg5: dance_b, 8192 envs, four environments per wavefront, 2 pipelined sub-batches)" \
    $(find /tmp/p_trace5 -name "*.db" | head -1) > /dev/null; head -12 $OUT/kstep_packed_summary.md
  timeout 300 python bench.py --workload rollout --steps 2048 --warmup 256 > $OUT/bench_rollout_fused.json 2> $OUT/bench_rollout_fused.err; cut -c1-200 $OUT/bench_rollout_fused.json
  # the horizon launch (dm_batch_rollout, k_rollout_packed): its own line with live PMC passes, the per-wave cycle spread, a kernel trace
  mkdir -p $OUT/raw_horizon
  DM_PROFILE_KEEP=$OUT/raw_horizon timeout 600 python bench.py --horizon-launch --no-cpu-baseline --no-gym-loop > $OUT/bench_cfg3_horizon_launch.json 2> $OUT/bench_cfg3_horizon_launch.err; cut -c1-200 $OUT/bench_cfg3_horizon_launch.json
  for rw in alive v3-config; do timeout 300 python bench.py --horizon-launch --reward $rw --no-pmc --no-cpu-baseline --no-gym-loop > $OUT/bench_cfg3_horizon_launch_$rw.json 2>/dev/null; cut -c1-160 $OUT/bench_cfg3_horizon_launch_$rw.json; done
  timeout 300 python bench.py --horizon-launch --workload cfg4 --no-pmc --no-cpu-baseline --no-gym-loop > $OUT/bench_cfg4_horizon_launch.json 2>/dev/null; cut -c1-160 $OUT/bench_cfg4_horizon_launch.json
  timeout 300 python bench.py --horizon-launch --workload cfg5 --no-pmc --no-cpu-baseline --no-gym-loop > $OUT/bench_cfg5_horizon_launch.json 2>/dev/null; cut -c1-160 $OUT/bench_cfg5_horizon_launch.json
  ( for a in "256 4096 imitation" "128 4096 imitation" "256 4096 alive" "128 4096 alive policy"; do timeout 120 python tools/horizon_wave_times.py $a | tail -1; done ) > $OUT/horizon_wave_times.txt 2>&1; cat $OUT/horizon_wave_times.txt
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/p_traceh -- python $OLDPWD/bench.py --horizon-launch --steps 512 --warmup 0 --prewarm-horizons 1 --no-pmc --no-cpu-baseline --no-gym-loop > /dev/null 2>&1 )
  ROWS=6 python tools/rocprof_summary.py $OUT/krollout_summary.md "horizon launch — $TAG, MI355X (bench.py --horizon-launch: cfg3 + 5-term imitation reward, 4096 envs, 256 steps per launch)" \
    $(find /tmp/p_traceh -name "*.db" | head -1) > /dev/null; head -12 $OUT/krollout_summary.md
  # driver-sized window as the driver runs it
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-gym-loop > $OUT/bench_cfg3_driver_window.json 2>/dev/null; cut -c1-160 $OUT/bench_cfg3_driver_window.json
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/p_trace -- python $OLDPWD/bench.py --steps 96 --warmup 16 --no-pmc --no-cpu-baseline --no-gym-loop > /dev/null 2>&1 )
  ROWS=6 python tools/rocprof_summary.py $OUT/kstep_summary.md "step kernel — $TAG, MI355X (bench.py default workload: cfg3 + 5-term imitation reward, 4096 envs as 2 pipelined sub-batches)" \
    $(find /tmp/p_trace -name "*.db" | head -1) > /dev/null; head -12 $OUT/kstep_summary.md
  timeout 400 python tools/train_trpo.py --envs 4096 --horizon 128 --seconds 25 --out $OUT/trpo_train.json 2>&1 | tail -2 | tee $OUT/trpo_train.log
  DM_TRPO_PROFILE=1 timeout 300 python tools/train_trpo.py --envs 4096 --horizon 128 --iters 40 --out $OUT/trpo_update_profile.json 2>&1 | tail -1
  timeout 120 python tools/vf_bench.py > $OUT/vf_bench.txt 2>&1; cat $OUT/vf_bench.txt
fi
