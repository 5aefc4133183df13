mkdir -p gpurun_out/r02i
timeout 1500 python tools/fuzz_parity.py 160 1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02i/fuzz.log | tail -20
