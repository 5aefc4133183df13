"""Generates tests/golden/trpo_update_golden.npz: inputs and float64 results of ONE TRPO update (src/trpo.py:235-296) computed by
the analytic numpy restatement tests/trpo_numpy.py.  Deterministic (seeded numpy only).  python tests/golden/gen/make_trpo_fixture.py"""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
sys.path.insert(0, ROOT)
from tests import trpo_numpy as TN  # noqa: E402


def normc(rng, a, b, std):
    w = rng.randn(a, b); return w * std / np.sqrt((w * w).sum(0, keepdims=True))


def main():
    rng = np.random.RandomState(20260927)
    T, N = 256, 2
    p = {"polfc1/w": normc(rng, 56, 100, 1.0), "polfc1/b": 0.05 * rng.randn(100), "polfc2/w": normc(rng, 100, 100, 1.0), "polfc2/b": 0.05 * rng.randn(100),
         "polfinal/w": normc(rng, 100, 28, 0.3), "polfinal/b": 0.02 * rng.randn(28), "logstd": -0.3 + 0.1 * rng.randn(1, 28),
         "vffc1/w": normc(rng, 56, 100, 1.0), "vffc1/b": np.zeros(100), "vffc2/w": normc(rng, 100, 100, 1.0), "vffc2/b": np.zeros(100),
         "vffinal/w": normc(rng, 100, 1, 1.0), "vffinal/b": np.zeros(1)}
    p = {k: v.astype(np.float32).astype(np.float64) for k, v in p.items()}          # exactly representable in the learner's float32
    rms = TN.Rms()
    scale = np.concatenate([0.3 * np.ones(28), 1.5 * np.ones(28)])
    rms.update(rng.randn(5000, 56) * scale + 0.05)                                # a filter that has seen data already
    rms0 = (rms.sum.copy(), rms.sumsq.copy(), rms.count)
    ob = (rng.randn(T, N, 56) * scale).astype(np.float32).astype(np.float64)
    mean, _ = TN.pol_forward(p, TN.obz(ob.reshape(-1, 56), rms))
    ac = (mean.reshape(T, N, 28) + np.exp(p["logstd"]) * rng.randn(T, N, 28)).astype(np.float32).astype(np.float64)
    rew = np.ones((T, N))
    new = (rng.rand(T, N) < 0.02).astype(np.int32); new[0] = 1
    vpred = (20 + 5 * rng.randn(T, N)).astype(np.float32).astype(np.float64)
    nextvpred = (20 + 5 * rng.randn(N)).astype(np.float32).astype(np.float64)
    # GAE per env (src/trpo.py:83-94), then env-major flattening (each worker's segment contiguous)
    adv = np.zeros((T, N)); ret = np.zeros((T, N))
    for e in range(N):
        nw = np.append(new[:, e], 0); vp = np.append(vpred[:, e], nextvpred[e]); last = 0.0
        for t in reversed(range(T)):
            nonterminal = 1 - nw[t + 1]
            delta = rew[t, e] + 0.995 * vp[t + 1] * nonterminal - vp[t]
            adv[t, e] = last = delta + 0.995 * 0.97 * nonterminal * last
        ret[:, e] = adv[:, e] + vpred[:, e]
    fl = lambda a: np.swapaxes(a, 0, 1).reshape((T * N,) + a.shape[2:])
    perms = [rng.permutation(T * N) for _ in range(3)]
    pnew, st = TN.update(p, rms, fl(ob), fl(ac), fl(adv), fl(ret), perms, vf_batch=128)
    out = {"T": T, "N": N, "ob": ob, "ac": ac, "rew": rew, "new": new, "vpred": vpred, "nextvpred": nextvpred, "perms": np.stack(perms),
           "rms0_sum": rms0[0], "rms0_sumsq": rms0[1], "rms0_count": rms0[2], "rms1_sum": rms.sum, "rms1_sumsq": rms.sumsq, "rms1_count": rms.count,
           "adv": adv, "tdlamret": ret}
    for k, v in p.items():
        out["p0/" + k] = v
    for k, v in pnew.items():
        out["p1/" + k] = v
    for k in ("g", "stepdir", "fullstep", "shs", "lm", "expectedimprove", "stepsize", "surrbefore", "surr", "kl"):
        out["st/" + k] = st[k]
    path = os.path.join(ROOT, "tests", "golden", "trpo_update_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "stepsize", st["stepsize"], "kl", st["kl"], "surr", st["surr"], "expectedimprove", st["expectedimprove"], "|g|", np.linalg.norm(st["g"]))


if __name__ == "__main__":
    main()
