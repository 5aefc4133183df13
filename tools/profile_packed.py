"""Diagnostic: per-stage shader-clock profile of k_step_packed (four environments per wavefront; DM options 105 + 101), one record
per WAVE, next to the one-env kernel's per-env profile of the same states."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from deepmimic_mujoco_amd import DPVecEnv, _abi as A

n = int(os.environ.get("DM_PROF_ENVS", "4096"))
full = os.environ.get("DM_PROF_WL", "cfg3") == "cfg3"
env = DPVecEnv(n, motion="walk", device=0, reward="alive", autoreset="rsi", seed=0, contacts=full, limits=full,
               action_mode="raw" if full else "p-control", frame_skip=1)
env.batch.set_option(105, 1)
env.reset("rsi")
rng = np.random.RandomState(0)
for t in range(40):
    env.step(rng.randn(n, 28) * (0.9 if full else 0.0))
env.batch.set_option(101, 1)
K = 5
recs = []
for t in range(K):
    env.step(rng.randn(n, 28) * (0.9 if full else 0.0))
    recs.append(env.batch.read_profile()[: (n + 3) // 4, :32].astype(np.float64))
p = np.concatenate(recs)
names = ["kinematics", "bias", "mass+factor", "rows", "constraint", "total"]
tot = p[:, 5]
print("k_step_packed, %d envs, cycles per WAVE-step (4 envs x 4 evaluations); mean total %.0f, median %.0f, p90 %.0f, max %.0f" % (n, tot.mean(), np.median(tot), np.percentile(tot, 90), tot.max()))
for k, nm in enumerate(names[:5]):
    print("   %-14s %9.0f (%.1f%%)" % (nm, p[:, k].mean(), 100 * p[:, k].mean() / tot.mean()))
print("   other          %9.0f" % (tot.mean() - p[:, :5].sum(1).mean()))
sub = ["row build", "imp + half solve + b", "A build", "warm start", "PGS", "assembly + L solve"]
for k, nm in enumerate(sub):
    print("      constraint/%-22s %9.0f" % (nm, p[:, 8 + k].mean()))
for k, nm in enumerate(["mass/f + M entries", "mass/elimination", "mass/D, scaling", "rows/geoms + limits", "rows/broad phase", "rows/narrow phase + emission"]):
    print("      %-32s %9.0f" % (nm, p[:, 16 + k].mean()))
print("      rows/largest candidate count of the wave per evaluation %.1f" % (p[:, 23].mean() / 4))
n2 = p[:, 7].sum()
if n2:
    print("      two-row-set evaluations: %.0f cycles each (PGS %.0f); one-row-set: %.0f each (PGS %.0f)" % (
        p[:, 24].sum() / n2, p[:, 22].sum() / n2, (p[:, 8:14].sum() - p[:, 24].sum()) / max(1, p[:, 15].sum() - n2), (p[:, 12].sum() - p[:, 22].sum()) / max(1, p[:, 15].sum() - n2)))
ev = np.maximum(p[:, 15], 1)
print("   constrained evaluations per wave-step %.2f of 4; mean wave nmax %.1f; two-row-set evaluations per wave-step %.3f; PGS loop trips per constrained evaluation %.1f" % (
    p[:, 15].mean(), (p[:, 14] / ev).mean(), p[:, 7].mean(), (p[:, 6] / ev).mean()))
print("   redo env-steps so far [total, candidates, box slots, contacts, rows, PGS test]:", env.batch.redo_reasons())
env.close()
