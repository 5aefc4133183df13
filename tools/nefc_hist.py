"""Constraint-row and contact counts of a STANDING population (noisy init pose, small actions, falls restart from the init pose) at the end of a few
horizons: how far above the packed path's 32 rows the environments are that a horizon launch re-steps in-wave.  python tools/nefc_hist.py (on the GPU box)"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from deepmimic_mujoco_amd import DPVecEnv, _abi as A
n, T = 4096, 64
env = DPVecEnv(n, motion="walk", device=0, reward="alive", autoreset="init", seed=0, packed=True, frame_skip=1)
b = env.batch; b.set_option(106, 1)
g = torch.Generator(device="cuda:0"); g.manual_seed(1)
ac = torch.randn((T + 1, n, 28), generator=g, dtype=torch.float64, device="cuda:0") * 0.1
ob = torch.zeros((T, n, 56), dtype=torch.float64, device="cuda:0"); rew = torch.zeros((T, n), dtype=torch.float64, device="cuda:0"); dn = torch.zeros((T, n), dtype=torch.uint8, device="cuda:0")
env.reset("init")
hist = np.zeros(80, dtype=np.int64); hc = np.zeros(40, dtype=np.int64)
for k in range(6):
    b.rollout(ac, (ob, rew, dn), 1); b.sync()
    ne = b.get(A.F_NEFC); nc = b.get(A.F_NCON)
    hist += np.bincount(np.minimum(ne, 79), minlength=80); hc += np.bincount(np.minimum(nc, 39), minlength=40)
print("nefc histogram (end of 6 horizons of 64 steps):", {i: int(v) for i, v in enumerate(hist) if v})
print("ncon histogram:", {i: int(v) for i, v in enumerate(hc) if v})
