"""What the policy step inside a horizon launch costs (round 6): the SAME physics trajectory twice — once with the policy in the loop (k_rollout_packed's policy_wave4 writes the
actions), once open loop with exactly those actions replayed — so that the difference of the two launches' durations is the in-wave policy + value step and nothing else.
    python tools/policy_cost.py [envs] [T] [repeats]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from deepmimic_mujoco_amd import DPVecEnv, MlpPolicy, _abi as A  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 256
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")
pol = MlpPolicy(device="cuda:0", seed=0)
w = pol.pack()
env = DPVecEnv(n, motion="walk", device=0, reward="alive", autoreset="rsi", seed=0, packed=True, frame_skip=1)
b = env.batch
b.set_option(106, 1)
obs = torch.zeros((T, n, 56), dtype=torch.float64, device=dev); rew = torch.zeros((T, n), dtype=torch.float64, device=dev); dn = torch.zeros((T, n), dtype=torch.uint8, device=dev)
vp = torch.zeros((T, n), dtype=torch.float32, device=dev)


def start():
    b.set_option(A.OPT_SEED, 0)
    env.reset("rsi")



def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); b.sync(); torch.cuda.synchronize()
    return time.perf_counter() - t0


res = {"policy": [], "replay": []}
for r in range(reps + 1):

    # a fresh env per repetition: identical seeds -> identical RSI draws -> identical trajectories
    e = DPVecEnv(n, motion="walk", device=0, reward="alive", autoreset="rsi", seed=0, packed=True, frame_skip=1)
    bb = e.batch; bb.set_option(106, 1); e.reset("rsi")
    q0 = bb.get(A.F_QPOS).copy(); v0 = bb.get(A.F_QVEL).copy(); f0 = bb.get(A.F_FRAME_IDX).copy()
    ac = torch.zeros((T + 1, n, 28), dtype=torch.float64, device=dev)
    t_pol = timed(lambda: bb.rollout(ac, (obs, rew, dn), 1, weights=w, vpred=vp, stochastic=True, seed=7, counter=0))
    obs_p = obs.clone(); dn_p = int(dn.sum())
    e.close()
    e = DPVecEnv(n, motion="walk", device=0, reward="alive", autoreset="rsi", seed=0, packed=True, frame_skip=1)
    bb = e.batch; bb.set_option(106, 1); e.reset("rsi")
    assert (bb.get(A.F_QPOS) == q0).all() and (bb.get(A.F_FRAME_IDX) == f0).all()
    t_rep = timed(lambda: bb.rollout(ac, (obs, rew, dn), 1))
    same = bool(torch.equal(obs, obs_p))
    e.close()
    if r:
        res["policy"].append(t_pol); res["replay"].append(t_rep)
    print("rep %d: policy in the loop %.2f ms, the same actions replayed %.2f ms, identical trajectories: %s, episodes ended %d" % (r, 1e3 * t_pol, 1e3 * t_rep, same, dn_p), flush=True)
tp = sorted(res["policy"])[len(res["policy"]) // 2]; tr = sorted(res["replay"])[len(res["replay"]) // 2]
print("%d envs x %d steps: with the policy %.2f M env-steps/s, replayed %.2f M; the policy + value step costs %.1f us per step = %.0f k cycles per wave-step at 2.4 GHz (%.1f %% of the launch)"
      % (n, T, n * T / tp / 1e6, n * T / tr / 1e6, 1e6 * (tp - tr) / T, 2.4e3 * 1e6 * (tp - tr) / T / 1e3, 100 * (tp - tr) / tp))
