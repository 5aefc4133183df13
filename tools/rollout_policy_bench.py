"""Rollout throughput (policy in the loop, segments + bookkeeping) under a GIVEN policy, horizon launch vs one fused launch per step.
Usage: python tools/rollout_policy_bench.py [shipped|untrained] [envs] [T]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from deepmimic_mujoco_amd import DPVecEnv, MlpPolicy, _abi as A  # noqa: E402
from deepmimic_mujoco_amd.rollout import SegmentCollector  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "shipped"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
T = int(sys.argv[3]) if len(sys.argv) > 3 else 256
CKPT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "ckpt", "trpo-walk-0")
for packed in (None, False):
    pol = MlpPolicy.from_tf_checkpoint(CKPT, device="cuda:0") if which == "shipped" else MlpPolicy(device="cuda:0", seed=0)
    pol.seed(1)
    env = DPVecEnv(n, motion="walk", device=0, reward="alive", autoreset="init", seed=0, packed=packed)
    env.batch.set_option(A.OPT_PIPELINE, 2)
    c = SegmentCollector(pol, env, T, stochastic=True, first_reset="init", fused=True)
    for _ in range(3):
        c.launch(); c.collect()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r0 = env.batch.redo_total()
    K = 4
    lens = []
    for _ in range(K):
        c.launch(); lens += c.collect()["ep_lens"]
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%s policy, %d envs, %d-step segments, %s: %.2f M env-steps/s; four envs per wave at the end: %s, kernel switches %d, env-steps re-stepped in the wave %.2e of all, "
          "episodes %d (mean length %.0f), max rows %d"
          % (which, n, T, "collector's choice" if packed is None else "one fused launch per step (one env per wave)", n * T * K / dt / 1e6, env.packed, c.kernel_switches,
             (env.batch.redo_total() - r0) / float(n * T * K), len(lens), sum(lens) / max(1, len(lens)), int(env.batch.get(A.F_NEFC).max())))
    env.close()
