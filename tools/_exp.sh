mkdir -p gpurun_out/r02q
for sd in $(seq 300 309); do timeout 300 python tools/fuzz_parity.py 320 $sd 2>&1 | grep -E "FAIL|fuzz seed" ; done | tee gpurun_out/r02q/fuzz.log
bash tools/gpu_round.sh r02q 2>&1 | cut -c1-200 | grep -E "passed|failed|smoke|value|own_algorithm" 
