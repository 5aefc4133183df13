#!/usr/bin/env python3
"""Which narrow-phase routines produce the contacts of a workload?  (exposure of the two pair types whose
routines are this repository's own algorithms, capsule-box and box-box, instead of restatements of MuJoCo's.)

    python tools/contact_exposure.py [--clip walk] [--envs 4096] [--steps 256] [--policy none|shipped]

Steps a batch with the benchmark's protocol (RSI auto-reset, N(0, 0.9^2) actions; or the shipped policy under the trainer's
episode protocol) with per-step diagnostics on, and histograms every contact of every step by its (geom type, geom type) pair."""
import argparse
import collections
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from deepmimic_mujoco_amd import DPVecEnv, _abi as A  # noqa: E402

TYPE = {0: "plane", 2: "sphere", 3: "capsule", 6: "box"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clip", default="walk"); ap.add_argument("--envs", type=int, default=4096); ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--policy", default="none", choices=["none", "shipped"])
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    n = args.envs
    shipped = args.policy == "shipped"
    env = DPVecEnv(n, motion=args.clip, device=0, reward="alive", autoreset="init" if shipped else "rsi", seed=0, diagnostics=True)
    gt = np.asarray(env._cm.geom_type)
    pol = None
    if shipped:
        import torch
        from deepmimic_mujoco_amd import MlpPolicy
        pol = MlpPolicy.from_tf_checkpoint(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "ckpt", "trpo-walk-0"), device="cuda:0")
        pol.seed(0)
    obs = env.reset("init" if shipped else "rsi")
    rng = np.random.RandomState(0)
    hist = collections.Counter(); rows = collections.Counter(); ncon_tot = 0; steps_with = collections.Counter()
    for t in range(args.steps):
        if shipped:
            import torch
            ac, _ = pol.act(True, torch.as_tensor(obs, device="cuda:0", dtype=torch.float64))
            a = ac.cpu().numpy()
        else:
            a = rng.randn(n, 28) * 0.9
        obs, rew, done, _ = env.step(a)
        cg = env.batch.get(A.F_CONTACT_GEOMS); ncon = env.batch.get(A.F_NCON); nefc = env.batch.get(A.F_NEFC)
        k = np.minimum(ncon, A.MAXEFC)
        mask = np.arange(A.MAXEFC)[None, :] < k[:, None]
        g1 = cg[:, :, 0][mask]; g2 = cg[:, :, 1][mask]
        for (a1, a2), c in zip(*np.unique(np.stack([gt[g1], gt[g2]], 1), axis=0, return_counts=True)):
            hist[(TYPE[int(a1)], TYPE[int(a2)])] += int(c)
        ncon_tot += int(k.sum())
        for v, c in zip(*np.unique(np.minimum(nefc, 63), return_counts=True)):
            rows[int(v)] += int(c)
        has_own = np.zeros(n, bool)
        own = ((gt[cg[:, :, 0]] == 3) & (gt[cg[:, :, 1]] == 6)) | ((gt[cg[:, :, 0]] == 6) & (gt[cg[:, :, 1]] == 6))
        has_own = (own & mask).any(1)
        steps_with["env-steps with a capsule-box or box-box contact"] += int(has_own.sum()); steps_with["env-steps"] += n
    res = {"clip": args.clip, "envs": n, "steps": args.steps, "policy": args.policy, "contacts": ncon_tot,
           "by_pair": {"%s-%s" % k: {"count": v, "frac": round(v / max(1, ncon_tot), 6)} for k, v in sorted(hist.items(), key=lambda kv: -kv[1])},
           "own_algorithm_contact_frac": round(sum(v for k, v in hist.items() if k in (("capsule", "box"), ("box", "box"))) / max(1, ncon_tot), 6),
           "env_steps_with_own_algorithm_contact_frac": round(steps_with["env-steps with a capsule-box or box-box contact"] / steps_with["env-steps"], 6),
           "nefc_hist": {str(k): v for k, v in sorted(rows.items())},
           "nefc_quantiles": {}}
    tot = sum(rows.values()); acc = 0
    for k in sorted(rows):
        acc += rows[k]
        for q in (0.5, 0.9, 0.99, 0.999):
            if str(q) not in res["nefc_quantiles"] and acc >= q * tot:
                res["nefc_quantiles"][str(q)] = k
    print(json.dumps(res, indent=1))
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)
    env.close()


if __name__ == "__main__":
    main()
