#!/bin/bash
# closed-loop per-step launches (bench.py --step-queue 0): kernel x batch size (x pipeline depth) — where DPVecEnv(packed=None)'s threshold belongs.  bash tools/ab_pipeline.sh <tag> "<envs...>" "<pipelines...>"
TAG=${1:-pipe}; mkdir -p gpurun_out/$TAG
Q="--no-pmc --no-cpu-baseline --no-gym-loop --no-vecenv-leg --no-horizon-leg --step-queue 0 --steps 512 --warmup 64 --repeats 3"
for n in ${2:-4096}; do for pl in ${3:-2}; do for pk in 0 1; do
  o=$(timeout 300 python bench.py $Q --envs $n --packed $pk --pipeline $pl 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('value %.3f M  spread %.3f..%.3f  kernel %s' % (j['value']/1e6, j['value_spread']['min']/1e6, j['value_spread']['max']/1e6, j['roofline']['kernel']))")
  echo "envs=$n packed=$pk pipeline=$pl : $o" | tee -a gpurun_out/$TAG/pipe.log
done; done; done
