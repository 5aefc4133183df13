"""Value-fit epoch timing: dm_vf_fit_epoch in its two forms on one update's worth of data (4096 envs x 128 steps, minibatches of 4096).
Usage: python tools/vf_bench.py  (under rocprofv3 --kernel-trace --stats for per-kernel durations)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from deepmimic_mujoco_amd.trpo import TrpoLearner  # noqa: E402
from deepmimic_mujoco_amd.policy import MlpPolicy  # noqa: E402

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = 4096 * 128
torch.manual_seed(0)
ob = torch.randn(n, 56, device="cuda:0"); ret = torch.randn(n, device="cuda:0")
for one in (False, True):
    pi = MlpPolicy(device="cuda:0", seed=5)
    L = TrpoLearner(pi, vf_batch_size=bs, vf_iters=3, vf_graph=False, vf_native=True)
    L.vf_epoch_filter = one
    inds = torch.randperm(n, device="cuda:0")
    L._vf_native_epoch(ob, ret, inds, bs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        L._vf_native_epoch(ob, ret, inds, bs)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("epoch_filter=%d: epoch of %d minibatches of %d: %.2f ms = %.1f us per minibatch" % (one, n // bs, bs, dt * 1e3, dt * 1e6 / (n // bs)))
