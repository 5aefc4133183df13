#!/bin/bash
# gpurun_out/<tag>/ (tools/gpu_round3.sh) -> profiles/r03_* : bash tools/copy_profiles.sh <tag>
set -e
O=gpurun_out/${1:-r03}; P=profiles
for f in cfg3 cfg3_alive cfg3_v3-config cfg3_driver_window cfg3_one_env_16384 cfg3_packed_16384 cfg3_packed_32768 cfg3_packed_4096 cfg2 rollout_fused cfg3_horizon_launch cfg3_horizon_launch_alive cfg3_horizon_launch_v3-config; do cp $O/bench_$f.json $P/r03_bench_$f.json; done
cp $O/bench_cfg4.json $P/r03_bench_cfg4_shard.json; cp $O/bench_cfg5.json $P/r03_bench_cfg5_shard.json; cp $O/bench_cfg5_no_reorder.json $P/r03_bench_cfg5_shard_no_reorder.json; cp $O/bench_cfg5_one_env_per_wave.json $P/r03_bench_cfg5_shard_one_env_per_wave.json
cp $O/bench_cfg4_horizon_launch.json $P/r03_bench_cfg4_shard_horizon_launch.json; cp $O/bench_cfg5_horizon_launch.json $P/r03_bench_cfg5_shard_horizon_launch.json
( cat $O/pytest_gpu.log; cat $O/smoke.log ) > $P/r03_gpu_tests.md
cp $O/stage_cycles.txt $P/r03_stage_cycles.md; cp $O/packed_stage_cycles.txt $P/r03_packed_stage_cycles.md; cp $O/ubench_lone.txt $P/r03_ubench_lone_wave.md
grep -v amdgpu.ids $O/horizon_wave_times.txt > $P/r03_horizon_wave_times.md; grep -v amdgpu.ids $O/vf_bench.txt > $P/r03_vf_bench.md
cp $O/trpo_update_profile.json $P/r03_trpo_update_profile_native.json; cp $O/trpo_train.json $P/r03_trpo_train_25s.json
for k in fetch_counters grbm_counters sq_counters write_counters sq_kernels trace_kernels; do cp $O/raw/$k.csv $P/raw/r03_cfg3_one_env_$k.csv; cp $O/raw_packed/$k.csv $P/raw/r03_cfg5_packed_$k.csv; cp $O/raw_horizon/$k.csv $P/raw/r03_cfg3_horizon_$k.csv; done
cp $O/kstep_summary.md $P/r03_kstep_summary.md; cp $O/kstep_packed_summary.md $P/r03_kstep_packed_summary.md; cp $O/krollout_summary.md $P/r03_krollout_summary.md
python tools/pmc_table.py $P/r03_kstep_summary.md profiles/raw/r03_cfg3_one_env > /dev/null
python tools/pmc_table.py $P/r03_kstep_packed_summary.md profiles/raw/r03_cfg5_packed > /dev/null
python tools/pmc_table.py $P/r03_krollout_summary.md profiles/raw/r03_cfg3_horizon "One launch = one 256-step horizon of all 4 096 environments (1 024 wavefronts): divide by 256 x 4 096 for per-env-step figures.  FETCH_SIZE / WRITE_SIZE (KiB) include the register save / restore of the step call per wave-step (1 616 B per lane of scratch: the callee uses the whole register file), about 120 MB written and 60 MB read per step - 0.7 TB/s, 9 % of the HBM peak." > /dev/null
python tools/kernel_resources.py > $P/r03_kernel_resources.md 2>/dev/null
