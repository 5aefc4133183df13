mkdir -p gpurun_out/e4
Q="--no-pmc --no-cpu-baseline --no-gym-loop --no-vecenv-leg --no-horizon-leg --step-queue 0 --steps 512 --warmup 64 --repeats 3"
for hq in 4 8 16; do for pl in 2 3 4 6 8; do
  o=$(GPU_MAX_HW_QUEUES=$hq timeout 300 python bench.py $Q --packed 1 --pipeline $pl 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('value %.3f M  spread %.3f..%.3f' % (j['value']/1e6, j['value_spread']['min']/1e6, j['value_spread']['max']/1e6))")
  echo "GPU_MAX_HW_QUEUES=$hq packed pipeline=$pl : $o" | tee -a gpurun_out/e4/hwq.log
done; done
