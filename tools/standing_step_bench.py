"""A synthetic standing population stepped CLOSED LOOP through per-step launches (`Batch.step` with device tensors, one launch set per call), by kernel:
OPT_PACKED 0 (one env per wave), 1 (four per wave, lean: 32 rows per env, overflows through the redo kernel), 2 (four per wave with the three-set code: 40 rows).
What `DPVecEnv(packed=None)`'s chooser decides between at >= 8 192 envs.   Usage: [DMENV_LIB=...] python tools/standing_step_bench.py [envs] [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from deepmimic_mujoco_amd import DPVecEnv, _abi as A  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
T = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = "cuda:0"
for mode in (0, 1, 2):
    env = DPVecEnv(n, motion="walk", device=0, reward="alive", autoreset="init", seed=0, packed=bool(mode), frame_skip=1)
    b = env.batch
    b.set_option(A.OPT_PACKED, mode); b.set_option(A.OPT_PIPELINE, 2)
    g = torch.Generator(device=dev); g.manual_seed(1)
    ac = torch.randn((T, n, 28), generator=g, dtype=torch.float64, device=dev) * 0.1
    ob = torch.zeros((n, 56), dtype=torch.float64, device=dev); rew = torch.zeros(n, dtype=torch.float64, device=dev); dn = torch.zeros(n, dtype=torch.uint8, device=dev)
    env.reset("init")
    for t in range(T):                                   # the population settles (and warms up)
        b.step(ac[t], 1, (ob, rew, dn))
    b.join(); b.sync()
    r0 = b.redo_total()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for rep in range(2):
        for t in range(T):
            b.step(ac[t], 1, (ob, rew, dn)); b.join()
    b.sync(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ne = b.get(A.F_NEFC)
    print("standing synthetic population, %d envs, closed loop, OPT_PACKED %d: %.3f M env-steps/s; re-stepped by the one-env code %.2f %% of env-steps; rows now: mean %.1f, above 32: %.1f %%"
          % (n, mode, n * T * 2 / dt / 1e6, 100.0 * (b.redo_total() - r0) / (n * T * 2), ne.mean(), 100.0 * (ne > 32).mean()))
    env.close()
