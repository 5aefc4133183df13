"""Per-stage shader-clock profile of the packed step INSIDE a horizon launch (k_rollout_packed), summed per wave over the horizon — the
counterpart of tools/profile_packed.py (one launch per step, dispatch order renewed every step).  Needs the diagnostic build:
    tools/build_variant.sh rprof -DDM_ROLLOUT_PROF ;  DMENV_LIB=build_ab/rprof.so python tools/profile_horizon.py [T] [envs]
DM_PROF_STANDING=1: a standing population (training's steady state) instead of the bench's falling one."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from deepmimic_mujoco_amd import DPVecEnv, _abi as A  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dev = "cuda:0"
STANDING = bool(os.environ.get("DM_PROF_STANDING"))      # a population that stands: noisy init pose, small actions, falls restart from the init pose
env = DPVecEnv(n, motion="walk", device=0, reward=os.environ.get("DM_PROF_REWARD", "alive"), autoreset="init" if STANDING else "rsi", seed=0, packed=True, frame_skip=1)
b = env.batch
b.set_option(106, 1)
g = torch.Generator(device=dev); g.manual_seed(1)
ac = torch.randn((T + 1, n, 28), generator=g, dtype=torch.float64, device=dev) * (0.1 if STANDING else 0.9)
ob = torch.zeros((T, n, 56), dtype=torch.float64, device=dev); rew = torch.zeros((T, n), dtype=torch.float64, device=dev)
dn = torch.zeros((T, n), dtype=torch.uint8, device=dev)
env.reset("init" if STANDING else "rsi")
for _ in range(2):
    b.rollout(ac, (ob, rew, dn), 1)
b.sync()
b.set_option(101, 1)
b.rollout(ac, (ob, rew, dn), 1)
b.sync()
nw = (n + 3) // 4
p = b.read_profile().reshape(-1)[: nw * 128].reshape(nw, 128)[:, :32].astype(np.float64) / T       # per wave, per step
tot = p[:, 5]
print("k_rollout_packed, %d envs, one %d-step horizon; cycles per WAVE-step (4 envs x 4 evaluations), averaged over the horizon; mean of the step bodies %.0f "
      "(whole horizon incl. re-steps and loop: %.0f per step), slowest wave %.0f" % (n, T, tot.mean(), p[:, 31].mean(), p[:, 31].max()))
for k, nm in enumerate(["kinematics", "bias", "mass+factor", "rows", "constraint"]):
    print("   %-14s %9.0f (%.1f%%)" % (nm, p[:, k].mean(), 100 * p[:, k].mean() / tot.mean()))
print("   other          %9.0f" % (tot.mean() - p[:, :5].sum(1).mean()))
for k, nm in enumerate(["row build", "imp + half solve + b", "A build", "warm start", "PGS", "assembly + L solve"]):
    print("      constraint/%-22s %9.0f" % (nm, p[:, 8 + k].mean()))
for k, nm in enumerate(["mass/f + M entries", "mass/elimination", "mass/D, scaling", "rows/geoms + limits", "rows/broad phase", "rows/narrow phase + emission"]):
    print("      %-32s %9.0f" % (nm, p[:, 16 + k].mean()))
n2 = p[:, 7].sum()
ev = np.maximum(p[:, 15], 1e-9)
if n2:
    print("      two-row-set evaluations: %.0f cycles each (PGS %.0f); one-row-set: %.0f each (PGS %.0f)" % (
        p[:, 24].sum() / n2, p[:, 22].sum() / n2, (p[:, 8:14].sum() - p[:, 24].sum()) / max(1e-9, p[:, 15].sum() - n2), (p[:, 12].sum() - p[:, 22].sum()) / max(1e-9, p[:, 15].sum() - n2)))
n3 = p[:, 25].sum()
if n3:
    print("      three-row-set evaluations (33 .. 40 rows somewhere in the wave): %.3f per wave-step, %.0f cycles each (PGS %.0f)" % (p[:, 25].mean(), p[:, 26].sum() / n3, p[:, 27].sum() / n3))
print("   constrained evaluations per wave-step %.2f of 4; mean wave nmax %.1f; two-row-set evaluations per wave-step %.3f; PGS loop trips per constrained evaluation %.1f" % (
    p[:, 15].mean(), (p[:, 14] / ev).mean(), p[:, 7].mean(), (p[:, 6] / ev).mean()))
print("   env-steps re-stepped in the wave so far [total, candidates, box slots, contacts, rows, PGS test]:", b.redo_reasons())
env.close()
