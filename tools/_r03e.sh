#!/bin/bash
mkdir -p gpurun_out/r03e; O=gpurun_out/r03e
Q="--no-cpu-baseline --no-pmc --no-gym-loop"
timeout 600 python -m pytest tests/test_gpu_rollout.py -q -x -k "horizon_launch" 2>&1 | tail -4 > $O/tests.txt
for c in 16 32 64 128 256; do timeout 300 python bench.py --steps 1024 $Q --horizon-launch --horizon-chunk $c > $O/cfg3_h$c.json 2> $O/cfg3_h.err; done
timeout 300 python bench.py --steps 1024 $Q --horizon-launch --no-reorder > $O/cfg3_hnoreorder.json 2>> $O/cfg3_h.err
timeout 300 python bench.py --workload rollout --steps 2048 > $O/rollout.json 2> $O/rollout.err
DM_TRPO_PROFILE=1 timeout 300 python tools/train_trpo.py --envs 4096 --horizon 128 --iters 25 --out $O/train.json > $O/train.log 2>&1
tail -2 $O/train.log
cat $O/tests.txt
for f in cfg3_h16 cfg3_h32 cfg3_h64 cfg3_h128 cfg3_h256 cfg3_hnoreorder rollout; do echo $f; python - <<PY
import json
try:
    d=json.loads(open("$O/$f.json").read().strip().splitlines()[-1])
    print(d["value"], d.get("ms_per_step"), d.get("horizon_launch"), d.get("packed_redo_env_steps"), d["config"].get("packed_redo_env_steps"))
except Exception as e: print("ERR", e)
PY
done
