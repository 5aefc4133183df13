set -u
mkdir -p gpurun_out/anchor
(timeout 1100 python tests/golden/gen/make_anchor.py 0 backend=gpu out=gpurun_out/anchor 2>&1 | tail -4) &
(timeout 1100 python tests/golden/gen/make_anchor.py 1 backend=gpu out=gpurun_out/anchor 2>&1 | tail -4) &
wait
