#!/bin/bash
# Collect the round's judged artifacts on the GPU box (run through gpurun from the repo root):
#   bench lines (cfg3 with cpu_baseline, cfg2, rollout), rocprofv3 kernel trace, PMC passes (HBM bytes, SQ mix).
# Writes gpurun_out/profile_<tag>/...; copy the summaries into profiles/ afterwards.
set -u
TAG=${1:-run}
OUT=$PWD/gpurun_out/profile_$TAG
mkdir -p $OUT
python bench.py > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err
python bench.py --workload cfg2 --no-cpu-baseline > $OUT/bench_cfg2.json 2>> $OUT/bench_cfg3.err
python bench.py --workload rollout --steps 1024 --warmup 256 > $OUT/bench_rollout.json 2>> $OUT/bench_cfg3.err
cd /tmp && export TMPDIR=/tmp
B="python $OLDPWD/bench.py --steps 96 --warmup 16 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d /tmp/p_trace -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p_fetch -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p_write -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY -d /tmp/p_sq -- $B > /dev/null 2>&1
cd $OLDPWD
ROWS=6 python tools/rocprof_summary.py $OUT/kstep_summary.md "k_step_narrow — $TAG, MI355X (bench.py default workload: cfg3 + 5-term imitation reward, 4096 envs)" \
  $(find /tmp/p_trace -name "*.db" | head -1) $(find /tmp/p_fetch -name "*.db" | head -1) $(find /tmp/p_write -name "*.db" | head -1) $(find /tmp/p_sq -name "*.db" | head -1) > /dev/null
cat $OUT/bench_cfg3.json | cut -c1-400
cat $OUT/bench_cfg2.json | cut -c1-120
cat $OUT/bench_rollout.json | cut -c1-160
tail -16 $OUT/kstep_summary.md
