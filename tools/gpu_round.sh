#!/bin/bash
# One gpurun call's worth of judged artefacts (run from the repo root on the GPU box): tests, bench lines, stage profile,
# contact exposure.  Outputs under gpurun_out/<tag>/ ; copy the summaries into profiles/ afterwards.
set -u
TAG=${1:-r02}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 ) > $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.log
timeout 600 python bench.py > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err; cut -c1-600 $OUT/bench_cfg3.json
# rocprofv3 --kernel-trace --stats of the same command (short run, no nested PMC passes), summarised per kernel
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/p_trace -- python $OLDPWD/bench.py --steps 96 --warmup 16 --no-pmc --no-cpu-baseline --no-gym-loop > /dev/null 2>&1 )
ROWS=6 python tools/rocprof_summary.py $OUT/kstep_summary.md "k_step_narrow — $TAG, MI355X (bench.py default workload: cfg3 + 5-term imitation reward, 4096 envs as 2 pipelined sub-batches)" \
  $(find /tmp/p_trace -name "*.db" | head -1) > /dev/null; head -12 $OUT/kstep_summary.md
for wl in cfg4 cfg5 cfg2; do
  timeout 300 python bench.py --workload $wl --no-pmc --no-cpu-baseline --no-gym-loop > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err; cut -c1-300 $OUT/bench_$wl.json
done
timeout 300 python bench.py --workload rollout --steps 2048 --warmup 256 > $OUT/bench_rollout_fused.json 2> $OUT/bench_rollout_fused.err; cut -c1-200 $OUT/bench_rollout_fused.json
for p in 1 2; do
  timeout 300 python bench.py --workload rollout --unfused --pipeline $p --steps 2048 --warmup 256 > $OUT/bench_rollout_p$p.json 2> $OUT/bench_rollout_p$p.err; cut -c1-200 $OUT/bench_rollout_p$p.json
done
timeout 300 python bench.py --dtype 32 --no-pmc --no-cpu-baseline --no-gym-loop > $OUT/bench_cfg3_f32.json 2> $OUT/bench_cfg3_f32.err; cut -c1-200 $OUT/bench_cfg3_f32.json
timeout 300 python bench.py --pipeline 1 --no-cpu-baseline --no-gym-loop > $OUT/bench_cfg3_pipeline1.json 2> $OUT/bench_cfg3_pipeline1.err; cut -c1-200 $OUT/bench_cfg3_pipeline1.json
DM_PROF_REWARD=imitation timeout 300 python tools/profile_stages.py > $OUT/stage_cycles.txt 2>&1; head -30 $OUT/stage_cycles.txt
timeout 300 python tools/contact_exposure.py --out $OUT/contact_exposure_cfg3.json > /dev/null 2> $OUT/contact_exposure.err
timeout 300 python tools/contact_exposure.py --policy shipped --envs 2048 --steps 400 --out $OUT/contact_exposure_policy.json > /dev/null 2>> $OUT/contact_exposure.err
grep -h "own_algorithm\|env_steps_with" $OUT/contact_exposure_*.json
# randomised HIP == oracle sweep (three seeds here; more: tools/fuzz_parity.py <states-per-clip> <seed>)
for sd in 1 2 3; do timeout 300 python tools/fuzz_parity.py 320 $sd 2>&1 | grep -E "FAIL|fuzz seed"; done | tee $OUT/fuzz.log
