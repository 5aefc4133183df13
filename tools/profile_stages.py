"""Diagnostic: per-stage shader-clock profile of k_step on the GPU (uses DM_OPT 101)."""
import os
import sys
import warnings
import numpy as np
warnings.simplefilter("ignore", UserWarning)   # frame_skip = 1 with the imitation reward: one mj_step per env step, as bench.py times it
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from deepmimic_mujoco_amd import DPVecEnv, _abi as A

for wl in ("cfg3", "cfg2"):
    full = wl == "cfg3"
    n = 4096
    env = DPVecEnv(n, motion="walk", device=0, reward=os.environ.get("DM_PROF_REWARD", "v3-config") if full else "alive", autoreset="rsi", seed=0,
                   contacts=full, limits=full, action_mode="raw" if full else "p-control", frame_skip=1)
    env.reset("rsi")
    rng = np.random.RandomState(0)
    for t in range(40):
        env.step(rng.randn(n, 28) * (0.9 if full else 0.0))
    env.batch.set_option(101, 1)
    acc = np.zeros(6); nefc = []; its = []
    K = 5
    for t in range(K):
        env.step(rng.randn(n, 28) * (0.9 if full else 0.0))
        p = env.batch.read_profile()
        acc += p[:, :6].mean(0); nefc.append(p[:, 6].mean()); its.append(p[:, 7].mean())
    acc /= K
    names = ["kinematics", "mass+factor", "bias(RNE)", "rows(collision)", "constraint", "total step"]
    print(wl, "mean cycles per env-step (4 evaluations):")
    for nme, v in zip(names, acc):
        print("   %-16s %10.0f  (%.1f%%)" % (nme, v, 100 * v / acc[5]))
    print("   mean nefc(last eval) %.2f  mean PGS sweeps %.2f" % (np.mean(nefc), np.mean(its)))
    p = env.batch.read_profile()
    sub = ["smooth solve", "J rows", "imp + half-solve", "A build", "warm start + PGS", "force assembly + back-solve"]
    for k, nme in enumerate(sub):  # slots 8..15
        print("      constraint/%-28s %9.0f" % (nme, p[:, 8 + k].mean()))
    rsub = ["geom poses + limit rows", "pass 0 broad phase", "pass 0 narrow phase", "pass 0 row emission", "pass 1 broad phase", "pass 1 narrow phase",
            "pass 1 row emission", "tail"]
    for k, nme in enumerate(rsub):  # slots 16..23
        print("      rows/%-34s %9.0f" % (nme, p[:, 16 + k].mean()))
    print("      rows/evaluations (of 4) with a pair past the bounding spheres: pass 0 %.2f, pass 1 %.2f; such pairs per evaluation: pass 0 %.2f, pass 1 %.2f"
          % (p[:, 24].mean(), p[:, 25].mean(), p[:, 26].mean() / 4, p[:, 27].mean() / 4))
    for lo, hi in [(0, 1), (1, 8), (8, 16), (16, 32), (32, 64)]:
        m = (p[:, 6] >= lo) & (p[:, 6] < hi)
        if m.any():
            print("   nefc [%d,%d): %5d envs, constraint %8.0f rows %8.0f total %8.0f | sweeps %.1f ybuild %6.0f half %6.0f A %6.0f ws %6.0f pgs %7.0f fin %6.0f" % (
                lo, hi, m.sum(), p[m, 4].mean(), p[m, 3].mean(), p[m, 5].mean(), p[m, 7].mean(), *[p[m, 8 + k].mean() for k in range(6)]))
            print("        rows parts: %s | evaluations with near pairs %.2f %.2f" % (" ".join("%6.0f" % p[m, 16 + k].mean() for k in range(8)), p[m, 24].mean(), p[m, 25].mean()))
    env.close()
