"""A synthetic STANDING population (noisy init pose, actions N(0, 0.1^2), falls restart from the init pose: 32 rows most of the time, 33 .. 37 on ~14 % of
its env-steps — three times the share of a trained policy's population) through 128-step horizon launches: env-steps/s and the share of env-steps the
packed path handed to the one-env code.  The same population as DM_PROF_STANDING=1 tools/profile_horizon.py, on the product library.
Usage: [DMENV_LIB=build_ab/X.so] python tools/standing_synth_bench.py [T] [envs]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from deepmimic_mujoco_amd import DPVecEnv, _abi as A  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dev = "cuda:0"
env = DPVecEnv(n, motion="walk", device=0, reward="alive", autoreset="init", seed=0, packed=True, frame_skip=1)
b = env.batch
b.set_option(106, 1)
g = torch.Generator(device=dev); g.manual_seed(1)
ac = torch.randn((T + 1, n, 28), generator=g, dtype=torch.float64, device=dev) * 0.1
ob = torch.zeros((T, n, 56), dtype=torch.float64, device=dev); rew = torch.zeros((T, n), dtype=torch.float64, device=dev)
dn = torch.zeros((T, n), dtype=torch.uint8, device=dev)
env.reset("init")
for _ in range(3):
    b.rollout(ac, (ob, rew, dn), 1)
b.sync()
r0 = b.redo_reasons()
K = 6
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(K):
    b.rollout(ac, (ob, rew, dn), 1)
b.sync(); torch.cuda.synchronize()
dt = time.perf_counter() - t0
r1 = b.redo_reasons()
ne = b.get(A.F_NEFC)
print("standing synthetic population, %d envs, %d x %d-step horizon launches: %.3f M env-steps/s; re-stepped by the one-env code %.2f %% of env-steps %s; rows now: mean %.1f, max %d, above 32: %.1f %%"
      % (n, K, T, n * T * K / dt / 1e6, 100.0 * (r1[0] - r0[0]) / (n * T * K), [x - y for x, y in zip(r1, r0)], ne.mean(), ne.max(), 100.0 * (ne > 32).mean()))
env.close()
