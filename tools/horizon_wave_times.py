"""Per-wave cycle totals of one horizon launch (dm_batch_rollout on the packed path, DM option 101): how long the horizon's slowest
wavefront takes against the mean — the part of the launch in which finished SIMDs idle.  Usage: python tools/horizon_wave_times.py [T] [envs]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from deepmimic_mujoco_amd import DPVecEnv, _abi as A  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
reward = sys.argv[3] if len(sys.argv) > 3 else "imitation"
with_policy = len(sys.argv) > 4 and sys.argv[4] == "policy"
dev = "cuda:0"
env = DPVecEnv(n, motion="walk", device=0, reward=reward, autoreset="rsi", seed=0, packed=True)
b = env.batch
g = torch.Generator(device=dev); g.manual_seed(1)
ac = torch.randn((T + 1, n, 28), generator=g, dtype=torch.float64, device=dev) * 0.9
ob = torch.zeros((T, n, 56), dtype=torch.float64, device=dev); rew = torch.zeros((T, n), dtype=torch.float64, device=dev)
dn = torch.zeros((T, n), dtype=torch.uint8, device=dev)
env.reset("rsi")
W = VP = None
if with_policy:
    from deepmimic_mujoco_amd import MlpPolicy
    pol = MlpPolicy(device=dev, seed=0); pol.pack()
    W = pol._packed; VP = torch.zeros((T, n), dtype=torch.float32, device=dev)
_roll = b.rollout
b.rollout = lambda a, o, k: _roll(a, o, k, W, VP, True, 7, 100)
for _ in range(2):
    b.rollout(ac, (ob, rew, dn), 1)
b.sync()
b.set_option(101, 1)
ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
for rep in range(3):
    ev0.record(torch.cuda.current_stream())
    b.rollout(ac, (ob, rew, dn), 1)
    ev1.record(torch.cuda.current_stream())
    b.sync(); torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1)
    w = b.read_profile()[: (n + 3) // 4, 5].astype(np.float64)
    ghz = w.max() / (ms * 1e-3) / 1e9
    print("horizon %d steps x %d envs (%s%s): launch %.2f ms = %.1f us / step -> %.2f M env-steps/s | per-wave cycles: mean %.3g, max %.3g (max / mean %.3f), "
          "std %.3g (%.1f %%), p99 %.3g | implied clock %.2f GHz | mean wave-step %.1f us"
          % (T, n, reward, ", policy in the loop" if with_policy else "", ms, ms / T * 1e3, n * T / ms / 1e3, w.mean(), w.max(), w.max() / w.mean(), w.std(), 100 * w.std() / w.mean(), np.percentile(w, 99), ghz,
             w.mean() / T / ghz / 1e3))
