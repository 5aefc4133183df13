"""float64 numpy restatement of ONE policy/value update of the reference's TRPO learner — TEST INFRASTRUCTURE.

Follows `learn()` of src/trpo.py line by line (`:235-296`) with the graph functions of `:104-163` written out by hand for
the reference's network (src/mlp_policy_trpo.py:24-61: obs filter with +-5 clip, two tanh layers of 100, linear head,
state-independent logstd; diagonal Gaussian of src/distributions.py:200-245).  TensorFlow builds these by automatic
differentiation; here every derivative is analytic (back-propagation and forward-mode products through the MLP), so this file
shares no code path with the torch-autograd learner it checks (deepmimic_mujoco_amd/trpo.py):

    surrogate gradient     g  = mean_n atarg_n * grad log pi(ac_n | ob_n)                                    (:126-134,161, at pi = oldpi)
    Fisher-vector product  F v = J^T Sigma^-1 J v / N_f  (mean parameters)  (+)  2 v  (logstd)               (:147-158: Hessian of mean KL(old || pi) at pi = oldpi,
                                                                                                              every 5th sample :245)
    CG 10 iterations, damping 0.1 (src/cg.py:2-34, :229), step = stepdir / sqrt(0.5 stepdir.F stepdir / max_kl) (:256-261),
    backtracking on the exact surrogate / KL (:262-283), value fit by MpiAdam (src/mpi_adam.py:21-35) over minibatches (:288-295).
"""
import numpy as np

POL = ("polfc1/w", "polfc1/b", "polfc2/w", "polfc2/b", "polfinal/w", "polfinal/b", "logstd")
VF = ("vffc1/w", "vffc1/b", "vffc2/w", "vffc2/b", "vffinal/w", "vffinal/b")


class Rms(object):
    """src/utils/misc_util.py:32-70"""

    def __init__(self, n=56, eps=1e-2):
        self.sum = np.zeros(n); self.sumsq = np.full(n, eps); self.count = eps

    def update(self, x):
        x = np.asarray(x, dtype=np.float64)
        self.sum += x.sum(0); self.sumsq += (x * x).sum(0); self.count += x.shape[0]

    @property
    def mean(self):
        return self.sum / self.count

    @property
    def std(self):
        return np.sqrt(np.maximum(self.sumsq / self.count - self.mean ** 2, 1e-2))


def flat(p, keys):
    return np.concatenate([np.asarray(p[k], dtype=np.float64).reshape(-1) for k in keys])


def unflat(theta, like, keys):
    out, o = {}, 0
    for k in keys:
        n = like[k].size
        out[k] = theta[o:o + n].reshape(like[k].shape); o += n
    return out


def obz(ob, rms):
    return np.clip((ob - rms.mean) / rms.std, -5.0, 5.0)


def pol_forward(p, x):
    h1 = np.tanh(x @ p["polfc1/w"] + p["polfc1/b"])
    h2 = np.tanh(h1 @ p["polfc2/w"] + p["polfc2/b"])
    return h2 @ p["polfinal/w"] + p["polfinal/b"], (x, h1, h2)


def pol_backward(p, cache, dmean, dlogstd):
    """gradient of sum(dmean * mean) + dlogstd . logstd with respect to the POL parameters (flat)."""
    x, h1, h2 = cache
    g = {}
    g["polfinal/w"] = h2.T @ dmean; g["polfinal/b"] = dmean.sum(0)
    dz2 = (dmean @ p["polfinal/w"].T) * (1 - h2 * h2)
    g["polfc2/w"] = h1.T @ dz2; g["polfc2/b"] = dz2.sum(0)
    dz1 = (dz2 @ p["polfc2/w"].T) * (1 - h1 * h1)
    g["polfc1/w"] = x.T @ dz1; g["polfc1/b"] = dz1.sum(0)
    g["logstd"] = np.asarray(dlogstd, dtype=np.float64).reshape(p["logstd"].shape)
    return flat(g, POL)


def pol_jvp(p, cache, v):
    """directional derivative of the mean along the parameter direction v (dict), forward mode."""
    x, h1, h2 = cache
    dz1 = x @ v["polfc1/w"] + v["polfc1/b"]
    dh1 = (1 - h1 * h1) * dz1
    dz2 = dh1 @ p["polfc2/w"] + h1 @ v["polfc2/w"] + v["polfc2/b"]
    dh2 = (1 - h2 * h2) * dz2
    return dh2 @ p["polfinal/w"] + h2 @ v["polfinal/w"] + v["polfinal/b"]


def vf_forward(p, x):
    h1 = np.tanh(x @ p["vffc1/w"] + p["vffc1/b"])
    h2 = np.tanh(h1 @ p["vffc2/w"] + p["vffc2/b"])
    return (h2 @ p["vffinal/w"] + p["vffinal/b"])[:, 0], (x, h1, h2)


def vf_backward(p, cache, dv):
    x, h1, h2 = cache
    dv = dv[:, None]
    g = {}
    g["vffinal/w"] = h2.T @ dv; g["vffinal/b"] = dv.sum(0)
    dz2 = (dv @ p["vffinal/w"].T) * (1 - h2 * h2)
    g["vffc2/w"] = h1.T @ dz2; g["vffc2/b"] = dz2.sum(0)
    dz1 = (dz2 @ p["vffc2/w"].T) * (1 - h1 * h1)
    g["vffc1/w"] = x.T @ dz1; g["vffc1/b"] = dz1.sum(0)
    return flat(g, VF)


def neglogp(ac, mean, logstd):
    return 0.5 * (((ac - mean) / np.exp(logstd)) ** 2).sum(-1) + 0.5 * np.log(2 * np.pi) * ac.shape[-1] + logstd.sum(-1)


def kl(mean0, logstd0, mean1, logstd1):
    return (logstd1 - logstd0 + (np.exp(2 * logstd0) + (mean0 - mean1) ** 2) / (2.0 * np.exp(2 * logstd1)) - 0.5).sum(-1)


def cg(f_Ax, b, cg_iters=10, residual_tol=1e-10):
    p = b.copy(); r = b.copy(); x = np.zeros_like(b); rdotr = r.dot(r)
    for _ in range(cg_iters):
        z = f_Ax(p)
        v = rdotr / p.dot(z)
        x += v * p; r -= v * z
        newrdotr = r.dot(r)
        p = r + (newrdotr / rdotr) * p
        rdotr = newrdotr
        if rdotr < residual_tol:
            break
    return x


def update(params, rms, ob, ac, adv, tdlamret, perms, max_kl=0.01, cg_iters=10, cg_damping=0.1, vf_stepsize=1e-3, vf_batch=128,
           fvp_subsample=5, adam=None):
    """One g-step of src/trpo.py:235-296 on a flat batch.  `params`: dict name -> float64 array (modified copy returned);
    `rms`: Rms (updated in place); `perms`: one index permutation per value-fit epoch; `adam`: dict(m, v, t) carried across calls.
    Returns (new params, stats dict)."""
    p = {k: np.array(v, dtype=np.float64) for k, v in params.items()}
    ob = np.asarray(ob, dtype=np.float64); ac = np.asarray(ac, dtype=np.float64)
    atarg = (adv - adv.mean()) / adv.std()                                    # :240
    rms.update(ob)                                                            # :242
    x = obz(ob, rms)
    old_mean, cache = pol_forward(p, x)                                       # :247 oldpi <- pi
    old_logstd = p["logstd"].copy()
    N = ob.shape[0]
    sig2 = np.exp(2 * old_logstd)

    def losses(q):
        mean, _ = pol_forward(q, x)
        ratio = np.exp(neglogp(ac, old_mean, old_logstd) - neglogp(ac, mean, q["logstd"]))
        return float((ratio * atarg).mean()), float(kl(old_mean, old_logstd, mean, q["logstd"]).mean())

    surrbefore, _ = losses(p)
    # gradient of the surrogate at pi = oldpi (ratio = 1): mean_n atarg_n grad log pi(ac_n)
    dmean = atarg[:, None] * (ac - old_mean) / sig2 / N
    dlogstd = (atarg[:, None] * ((ac - old_mean) ** 2 / sig2 - 1.0)).sum(0) / N
    g = pol_backward(p, cache, dmean, dlogstd)
    xs = x[::fvp_subsample]
    _, cache_f = pol_forward(p, xs)
    Nf = xs.shape[0]

    def fvp(vflat):
        v = unflat(vflat, p, POL)
        jv = pol_jvp(p, cache_f, v)
        return pol_backward(p, cache_f, jv / sig2 / Nf, 2.0 * v["logstd"].reshape(-1)) + cg_damping * vflat     # :229

    stepdir = cg(fvp, g, cg_iters)
    shs = 0.5 * stepdir.dot(fvp(stepdir))
    lm = np.sqrt(shs / max_kl)
    fullstep = stepdir / lm
    expectedimprove = float(g.dot(fullstep))
    thbefore = flat(p, POL)
    stepsize, ok = 1.0, False
    for _ in range(10):                                                       # :266-283
        q = dict(p); q.update(unflat(thbefore + fullstep * stepsize, p, POL))
        surr, klv = losses(q)
        improve = surr - surrbefore
        if np.isfinite(surr) and np.isfinite(klv) and klv <= max_kl * 1.5 and improve >= 0:
            ok = True
            break
        stepsize *= 0.5
    if ok:
        p.update(unflat(thbefore + fullstep * stepsize, p, POL))
    else:
        surr, klv = losses(p)
    # value function :288-295
    if adam is None:
        adam = {"m": np.zeros(flat(p, VF).size), "v": np.zeros(flat(p, VF).size), "t": 0}
    for perm in perms:
        for o in range(0, N - vf_batch + 1, vf_batch):
            mb = perm[o:o + vf_batch]
            rms.update(ob[mb])                                                # :293
            vp, c = vf_forward(p, obz(ob[mb], rms))
            gv = vf_backward(p, c, 2.0 * (vp - tdlamret[mb]) / len(mb))
            adam["t"] += 1
            a = vf_stepsize * np.sqrt(1 - 0.999 ** adam["t"]) / (1 - 0.9 ** adam["t"])
            adam["m"] = 0.9 * adam["m"] + 0.1 * gv
            adam["v"] = 0.999 * adam["v"] + 0.001 * gv * gv
            th = flat(p, VF) - a * adam["m"] / (np.sqrt(adam["v"]) + 1e-8)
            p.update(unflat(th, p, VF))
    stats = {"g": g, "stepdir": stepdir, "shs": float(shs), "lm": float(lm), "fullstep": fullstep, "expectedimprove": expectedimprove,
             "stepsize": stepsize if ok else 0.0, "surrbefore": surrbefore, "surr": surr, "kl": klv, "atarg": atarg}
    return p, stats
