# learner bundles on the GPU box:  [SHARES="1 0"] bash tools/prof_train.sh <tag> [ubench] [tests] [train] [vfbatch] [trace]
#   train: 80 iterations + update phase times per DM_VF_SHARE setting;  trace: kernel trace of 20 iterations -> train_kernels.md, idle gaps, kernel spans
TAG=$1; shift
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
for WHAT in "$@"; do case $WHAT in
  ubench) build_ab/mfma_f32_gemm | tee gpurun_out/$TAG/mfma_layout.txt ;;
  tests) ( timeout 900 python -m pytest tests/test_trpo.py tests/test_gpu_rollout.py -x -q -m gpu 2>&1 | tail -15 ) | tee gpurun_out/$TAG/pytest.log ;;
  train)
    for sh in ${SHARES:-0 1}; do
      DM_VF_SHARE=$sh timeout 300 python tools/train_trpo.py --envs 4096 --horizon 128 --iters 80 --out gpurun_out/$TAG/train80_share$sh.json 2>&1 | tail -1 | sed "s/^/share=$sh /"
      DM_VF_SHARE=$sh DM_TRPO_PROFILE=1 timeout 300 python tools/train_trpo.py --envs 4096 --horizon 128 --iters 40 --out gpurun_out/$TAG/prof40_share$sh.json 2>&1 | tail -1
      python - <<PY
import json
h=json.load(open("gpurun_out/$TAG/prof40_share$sh.json"))["history"][10:]
keys=h[0]["profile_ms"].keys()
print("share=$sh", {k: round(sum(x["profile_ms"][k] for x in h)/len(h),2) for k in keys})
PY
    done ;;
  vfbatch)
    for vb in ${VFB:-4096 8192 16384}; do
      timeout 300 python tools/train_trpo.py --envs 4096 --horizon 128 --iters 120 --vf-batch $vb --out gpurun_out/$TAG/train120_vfb$vb.json 2>&1 | tail -1 | sed "s/^/vf_batch=$vb /"
    done ;;
  trace)
    ( cd /tmp && DM_VF_SHARE=${SHARE:-0} rocprofv3 --kernel-trace --stats -d /tmp/p_train -- python $OLDPWD/tools/train_trpo.py --envs 4096 --horizon 128 --iters 20 --out /tmp/t.json > /dev/null 2>&1 )
    ROWS=9 python tools/rocprof_summary.py gpurun_out/$TAG/train_kernels.md "training loop, 20 iterations (4096 envs x 128 steps) — MI355X" $(find /tmp/p_train -name "*.db" | head -1) | head -16
    python tools/gpu_gaps.py $(find /tmp/p_train -name "*.db" | head -1) | head -3 | tee gpurun_out/$TAG/gaps.txt
    python tools/gpu_span.py $(find /tmp/p_train -name "*.db" | head -1) k_rollout_packed "k_pg<1>" > gpurun_out/$TAG/span_rollout_to_grad.txt
    python tools/gpu_span.py $(find /tmp/p_train -name "*.db" | head -1) "k_pg<0>" k_fill_rows > gpurun_out/$TAG/span_linesearch_to_rollout.txt ;;
esac; done
