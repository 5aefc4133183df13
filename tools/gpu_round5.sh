#!/bin/bash
# Round-5 gpurun bundles (same bundles as round 4) (run from the repo root on the GPU box):  bash tools/gpu_round5.sh <tag> <what...>
#   what: newtests | tests | ab "<variants>" | stage | bench | standing | train
# Outputs under gpurun_out/<tag>/ ; summaries worth keeping are copied into profiles/ by hand.
set -u
TAG=${1:-r05}; shift || true
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
LIB=deepmimic_mujoco_amd/csrc/libdmenv.so
Q="--no-pmc --no-cpu-baseline --no-gym-loop"
for WHAT in "$@"; do case "$WHAT" in
  newtests)
    ( timeout 900 python -m pytest tests/test_gpu_queue.py -x -q 2>&1 | tail -15 ) | tee $OUT/pytest_new.log ;;
  tests)
    ( timeout 2400 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -60 ) > $OUT/pytest_gpu.log; tail -14 $OUT/pytest_gpu.log
    python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.log ;;
  ab:*)
    # A/B of whole-library builds under build_ab/ inside one call: the driver's 20-step window and the 512-step default, alternating
    for rep in 1 2; do for v in ${WHAT#ab:}; do
      for st in "20 5" "512 64"; do set -- $st
        o=$(DMENV_LIB=$PWD/build_ab/$v.so timeout 300 python bench.py $Q --steps $1 --warmup $2 2>$OUT/ab_err.txt | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('value %.3f M  spread %.2f..%.2f  vecenv %s  horizon %s  launch_us %s' % (j['value']/1e6, j['value_spread']['min']/1e6, j['value_spread']['max']/1e6,
      j['vecenv_step'] and round(j['vecenv_step']['value']/1e6,3), j['horizon_launch'] and round(j['horizon_launch']['value']/1e6,3), j['roofline']['launch'].get('avg_us')))" 2>>$OUT/ab_err.txt)
        echo "$v steps=$1 : ${o:-FAILED $(tail -2 $OUT/ab_err.txt | cut -c1-200)}" | tee -a $OUT/ab.log
      done
    done; done ;;
  hl:*)
    # horizon-launch leg only, five windows of 1024 steps, imitation and alive rewards: tight A/B of builds whose difference is a per-cent
    for rep in 1 2; do for v in ${WHAT#hl:}; do for rw in ${HL_REWARDS:-imitation alive}; do
      o=$(DMENV_LIB=$PWD/build_ab/$v.so timeout 300 python bench.py $Q --no-vecenv-leg --no-horizon-leg --reward $rw --steps 1024 --warmup 0 --repeats 5 2>$OUT/hl_err.txt | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('value %.3f M  spread %.3f..%.3f  launch_us %s' % (j['value']/1e6, j['value_spread']['min']/1e6, j['value_spread']['max']/1e6, j['roofline']['launch'].get('avg_us')))")
      echo "$v $rw : ${o:-FAILED $(tail -2 $OUT/hl_err.txt | cut -c1-200)}" | tee -a $OUT/hl.log
    done; done; done ;;
  ps:*)
    # one launch set per call (k_step_narrow at 4096 envs, k_step_packed at 8192): tight A/B of builds, five windows of 512 steps
    for rep in 1 2; do for v in ${WHAT#ps:}; do for wl in cfg3 cfg5; do
      o=$(DMENV_LIB=$PWD/build_ab/$v.so timeout 300 python bench.py $Q --no-vecenv-leg --no-horizon-leg --step-queue 0 --workload $wl --steps 512 --warmup 64 --repeats 5 2>$OUT/ps_err.txt | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('value %.3f M  spread %.3f..%.3f  kernel %s' % (j['value']/1e6, j['value_spread']['min']/1e6, j['value_spread']['max']/1e6, j['roofline']['kernel']))")
      echo "$v $wl : ${o:-FAILED $(tail -2 $OUT/ps_err.txt | cut -c1-200)}" | tee -a $OUT/ps.log
    done; done; done ;;
  pol:*)
    # the policy step inside the horizon launch: rollout workload (untrained policy + GAE) and standing workload (shipped policy), per build
    for rep in 1 2; do for v in ${WHAT#pol:}; do
      o=$(DMENV_LIB=$PWD/build_ab/$v.so timeout 300 python bench.py --workload rollout --steps 2048 --warmup 256 2>$OUT/pol_err.txt | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('rollout %.3f M' % (j['value']/1e6))")
      o2=$(DMENV_LIB=$PWD/build_ab/$v.so timeout 300 python bench.py --workload standing --steps 1024 --warmup 768 2>>$OUT/pol_err.txt | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('standing packed %.3f M  one-env %.3f M' % (j['legs']['packed']['value']/1e6, j['legs']['one_env']['value']/1e6))")
      echo "$v : ${o:-FAILED} ; ${o2:-FAILED $(tail -2 $OUT/pol_err.txt | cut -c1-200)}" | tee -a $OUT/pol.log
    done; done ;;
  stage:*)
    for v in ${WHAT#stage:}; do echo "== $v" | tee -a $OUT/stage.log; DMENV_LIB=$PWD/build_ab/$v.so timeout 300 python tools/profile_packed.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/stage.log; done ;;
  hstage:*)
    for v in ${WHAT#hstage:}; do echo "== $v" | tee -a $OUT/hstage.log; DMENV_LIB=$PWD/build_ab/$v.so timeout 300 python tools/profile_horizon.py ${HSTAGE_ARGS:-} 2>&1 | grep -v amdgpu.ids | tee -a $OUT/hstage.log; done ;;
  train)
    timeout 400 python tools/train_trpo.py --envs 4096 --horizon 128 --seconds 60 --out $OUT/trpo_train_60s.json 2>&1 | tail -3 | tee $OUT/trpo_train.log
    DM_TRPO_PROFILE=1 timeout 300 python tools/train_trpo.py --envs 4096 --horizon 128 --iters 40 --out $OUT/trpo_update_profile.json 2>&1 | tail -1
    timeout 400 python tools/train_trpo.py --envs 4096 --horizon 128 --seconds 60 --reward imitation --frame-skip mocap --out $OUT/trpo_imitation_60s.json 2>&1 | tail -2 | tee $OUT/trpo_imitation.log ;;
  standing)
    timeout 600 python bench.py --workload standing --steps 2048 --warmup 1024 > $OUT/bench_standing.json 2> $OUT/bench_standing.err; cut -c1-1500 $OUT/bench_standing.json; tail -3 $OUT/bench_standing.err ;;
  pytest:*)
    ( timeout 1500 python -m pytest ${WHAT#pytest:} -x -q 2>&1 | tail -25 ) | tee -a $OUT/pytest_sel.log ;;
  bench)
    rm -rf $OUT/rawq $OUT/raw1; mkdir -p $OUT/raw
    DM_PROFILE_KEEP=$OUT/rawq timeout 900 python bench.py > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err; cut -c1-600 $OUT/bench_cfg3.json
    for f in $OUT/rawq/*; do mv $f $OUT/raw/${TAGP:-r05}_cfg3_queue_$(basename $f); done
    # the one-launch-set-per-call form with its own counter passes (what vecenv_step times): raw exports for the k_step_narrow summary
    DM_PROFILE_KEEP=$OUT/raw1 timeout 600 python bench.py --step-queue 0 --no-cpu-baseline --no-gym-loop --no-vecenv-leg --no-horizon-leg > $OUT/bench_cfg3_unqueued_pmc.json 2>/dev/null; cut -c1-200 $OUT/bench_cfg3_unqueued_pmc.json
    for f in $OUT/raw1/*; do mv $f $OUT/raw/${TAGP:-r05}_cfg3_kstep_$(basename $f); done
    timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $Q > $OUT/bench_cfg3_driver_window.json 2>/dev/null; cut -c1-300 $OUT/bench_cfg3_driver_window.json ;;
  benchall)
    for rw in alive v3-config; do timeout 300 python bench.py --reward $rw $Q > $OUT/bench_cfg3_$rw.json 2>/dev/null; cut -c1-200 $OUT/bench_cfg3_$rw.json; done
    for wl in cfg2 cfg4 cfg5; do timeout 300 python bench.py --workload $wl $Q > $OUT/bench_$wl.json 2>/dev/null; cut -c1-200 $OUT/bench_$wl.json; done
    timeout 300 python bench.py --step-queue 0 $Q > $OUT/bench_cfg3_unqueued.json 2>/dev/null; cut -c1-200 $OUT/bench_cfg3_unqueued.json
    timeout 300 python bench.py --workload rollout --steps 2048 --warmup 256 > $OUT/bench_rollout_fused.json 2>/dev/null; cut -c1-200 $OUT/bench_rollout_fused.json ;;
  trace)
    # the judged command's kernel trace: the profiled child runs nothing but horizons of 256 queued steps (one untimed + the 512 timed ones = 3 launches)
    ( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/p_trace -- python $OLDPWD/bench.py --_child --steps 512 --warmup 0 --prewarm-horizons 1 $Q > /dev/null 2>&1 )
    ROWS=8 python tools/rocprof_summary.py $OUT/krollout_summary.md "step queue -> horizon launches — $TAG, MI355X (bench.py default: cfg3 + 5-term imitation reward, 4096 envs, dm_batch_step calls queued 256 per launch; 3 launches)" \
      $(find /tmp/p_trace -name "*.db" | head -1) > /dev/null; head -14 $OUT/krollout_summary.md
    ( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/p_trace1 -- python $OLDPWD/bench.py --_child --step-queue 0 --steps 96 --warmup 16 --prewarm-horizons 1 $Q > /dev/null 2>&1 )
    ROWS=6 python tools/rocprof_summary.py $OUT/kstep_summary.md "one launch set per call — $TAG, MI355X (bench.py --step-queue 0: the kernel DPVecEnv picks at 4096 envs — k_step_packed + k_step_redo since round 6 — as 2 pipelined sub-batches; what vecenv_step times)" \
      $(find /tmp/p_trace1 -name "*.db" | head -1) > /dev/null; head -10 $OUT/kstep_summary.md
    # the PMC tables of both summaries from the raw per-launch counter exports the `bench` bundle kept (run `bench` first)
    ls $OUT/raw/*_cfg3_queue_*_counters.csv > /dev/null 2>&1 && python tools/pmc_table.py $OUT/krollout_summary.md $OUT/raw/${TAGP:-r05}_cfg3_queue > /dev/null
    ls $OUT/raw/*_cfg3_kstep_*_counters.csv > /dev/null 2>&1 && python tools/pmc_table.py $OUT/kstep_summary.md $OUT/raw/${TAGP:-r05}_cfg3_kstep > /dev/null
    tail -12 $OUT/krollout_summary.md ;;
  *) echo "unknown bundle $WHAT" ;;
esac; done
