#!/usr/bin/env python3
"""Generate tests/golden/learner_ref_golden.npz by EXECUTING the reference's own learner helpers on seeded inputs.

Runs only inside the build container (needs /root/reference).  Nothing of the reference is copied into the repo: its modules are imported from
where they lie — or, where a module drags in TensorFlow / mpi4py / gym at import time, the ONE function or class under test is cut out of the
module's syntax tree in memory and executed against the small stand-ins below — and only input / output DATA is written.

  imported as they are (numpy / scipy only):
    cg                          src/cg.py:2-34
    iterbatches                 src/dataset.py:50-60          (the value fit's minibatch order: np.random.shuffle + array_split)
    explained_variance, discount  src/utils/math_util.py:5-36
  executed from the syntax tree (their modules import tensorflow / mpi4py):
    add_vtarg_and_adv           src/trpo.py:83-94             (GAE(lambda) over a segment)
    MpiAdam.__init__ / update   src/mpi_adam.py:6-35          (stand-ins: a one-rank communicator, a numpy-backed U.GetFlat / SetFromFlat / numel)
    RunningMeanStd.update and the two expressions that define .mean / .std
                                src/utils/misc_util.py:32-70  (stand-ins: eager numpy versions of the five tf calls the two expressions make)

What the repo's own code (deepmimic_mujoco_amd/trpo.py cg / MpiAdam / explained_variance, rollout.add_vtarg_and_adv, policy.RunningMeanStd, the device
kernels dm_gae / dm_rms_update / dm_vf_fit_epoch's Adam rule) is then held to: tests/test_trpo.py, tests/test_policy.py, tests/test_gpu_rollout.py.
"""
import ast
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", "..", ".."))
REF = "/root/reference/src"
OUT = os.path.join(REPO, "tests", "golden", "learner_ref_golden.npz")


def cut(path, name, kind):
    """the source of ONE top-level def / class of a reference module, compiled on its own (never written anywhere)"""
    src = open(path).read()
    tree = ast.parse(src)
    for node in tree.body:
        if isinstance(node, kind) and node.name == name:
            mod = ast.Module(body=[node], type_ignores=[])
            return compile(mod, path, "exec"), node
    raise KeyError(name)


# ---- stand-ins -----------------------------------------------------------------------------------------------------------------
class OneRank(object):                       # mpi4py's communicator for a single worker
    def Allreduce(self, src, dst, op=None):
        dst[...] = src

    def Get_size(self):
        return 1

    def Get_rank(self):
        return 0

    def Bcast(self, buf, root=0):
        pass


class FlatVars(object):                      # utils.tf_util.GetFlat / SetFromFlat / numel over a list of numpy arrays
    @staticmethod
    def numel(v):
        return int(np.prod(v.shape))

    class GetFlat(object):
        def __init__(self, var_list):
            self.vl = var_list

        def __call__(self):
            return np.concatenate([v.ravel() for v in self.vl]).astype(np.float32)

    class SetFromFlat(object):
        def __init__(self, var_list):
            self.vl = var_list

        def __call__(self, theta):
            i = 0
            for v in self.vl:
                n = v.size
                v[...] = np.asarray(theta[i:i + n], dtype=v.dtype).reshape(v.shape)
                i += n


class EagerTf(object):                       # the five TensorFlow calls of RunningMeanStd's mean / std expressions, eager, float32 where TF is
    @staticmethod
    def to_float(x):
        return np.asarray(x).astype(np.float32)

    @staticmethod
    def sqrt(x):
        return np.sqrt(np.asarray(x, dtype=np.float32))

    @staticmethod
    def maximum(a, b):
        return np.maximum(np.asarray(a, dtype=np.float32), np.float32(b))

    @staticmethod
    def square(x):
        return np.square(np.asarray(x, dtype=np.float32))


def main():
    sys.path.insert(0, REF)
    out = {}
    rng = np.random.RandomState(20240928)

    # ---- cg -----------------------------------------------------------------------------------------------------------------
    from cg import cg
    n = 40
    Bm = rng.randn(n, n)
    A = Bm @ Bm.T / n + 0.1 * np.eye(n)
    b = rng.randn(n)
    out["cg_A"], out["cg_b"] = A, b
    for iters in (1, 3, 10, 60):
        out["cg_x_%d" % iters] = cg(lambda p: A @ p, b.copy(), cg_iters=iters)
    out["cg_x_tol"] = cg(lambda p: A @ p, b.copy(), cg_iters=60, residual_tol=1e-3)          # the early break
    A32 = A.astype(np.float32); b32 = b.astype(np.float32)
    out["cg_x32_10"] = cg(lambda p: A32 @ p, b32.copy(), cg_iters=10)                        # as the trainer calls it (float32 vectors)

    # ---- iterbatches (the value fit's minibatches: src/trpo.py:288-296) ---------------------------------------------------------
    import dataset
    N, bs = 1000, 128
    idx = np.arange(N)
    np.random.seed(77)
    order = []
    for _ in range(3):                                            # vf_iters epochs, one shuffle each
        for (mb,) in dataset.iterbatches((idx,), include_final_partial_batch=False, batch_size=bs):
            order.append(mb.copy())
    out["iterbatches_order"] = np.stack(order)                    # [3 * 7, 128]
    out["iterbatches_seed"] = np.array([77, N, bs])

    # ---- math_util -----------------------------------------------------------------------------------------------------------
    from utils import math_util
    y = rng.randn(500).astype(np.float32) * 3 + 1
    yp = (y + rng.randn(500).astype(np.float32)).astype(np.float32)
    out["ev_y"], out["ev_ypred"] = y, yp
    out["ev"] = np.array([math_util.explained_variance(yp, y), math_util.explained_variance(np.zeros_like(y), y), math_util.explained_variance(yp, np.ones_like(y))])
    x = rng.randn(64, 3)
    out["discount_x"], out["discount_y"] = x, math_util.discount(x, 0.995)

    # ---- add_vtarg_and_adv ----------------------------------------------------------------------------------------------------
    code, _ = cut(os.path.join(REF, "trpo.py"), "add_vtarg_and_adv", ast.FunctionDef)
    ns = {"np": np}
    exec(code, ns)
    T, E = 256, 6
    rew = rng.rand(T, E).astype(np.float32)
    vpred = (rng.randn(T, E) * 5).astype(np.float32)
    new = (rng.rand(T, E) < 0.04).astype(np.int32); new[0] = 1
    new[:, 1] = 0; new[0, 1] = 1                                  # one env that never resets inside the segment
    nxt = (rng.randn(E) * 5).astype(np.float32); nxt[2] = 0.0     # (nextvpred is zeroed by the generator when the segment ends an episode)
    adv = np.zeros((T, E), np.float32); ret = np.zeros((T, E), np.float32)
    for e in range(E):
        seg = {"new": new[:, e].copy(), "vpred": vpred[:, e].copy(), "nextvpred": nxt[e], "rew": rew[:, e].copy()}
        ns["add_vtarg_and_adv"](seg, 0.995, 0.97)
        adv[:, e], ret[:, e] = seg["adv"], seg["tdlamret"]
    out.update(gae_rew=rew, gae_vpred=vpred, gae_new=new, gae_nextvpred=nxt, gae_adv=adv, gae_tdlamret=ret, gae_gamma_lam=np.array([0.995, 0.97]))

    # ---- MpiAdam -------------------------------------------------------------------------------------------------------------
    code, _ = cut(os.path.join(REF, "mpi_adam.py"), "MpiAdam", ast.ClassDef)
    ns = {"np": np, "U": FlatVars, "MPI": types.SimpleNamespace(COMM_WORLD=OneRank(), SUM=None)}
    exec(code, ns)
    w = (rng.randn(7, 5) * 0.3).astype(np.float32); bb = (rng.randn(5) * 0.1).astype(np.float32)
    out["adam_theta0"] = np.concatenate([w.ravel(), bb.ravel()])
    opt = ns["MpiAdam"]([w, bb], epsilon=1e-8)
    grads = (rng.randn(12, 40) * np.logspace(-3, 1, 12)[:, None]).astype(np.float32)
    traj = []
    for g in grads:
        opt.update(g, 1e-3)
        traj.append(np.concatenate([w.ravel(), bb.ravel()]).copy())
    out["adam_grads"], out["adam_theta"] = grads, np.stack(traj)
    out["adam_m"], out["adam_v"] = opt.m.copy(), opt.v.copy()

    # ---- RunningMeanStd --------------------------------------------------------------------------------------------------------
    _code, node = cut(os.path.join(REF, "utils", "misc_util.py"), "RunningMeanStd", ast.ClassDef)
    init = [f for f in node.body if isinstance(f, ast.FunctionDef) and f.name == "__init__"][0]
    upd = [f for f in node.body if isinstance(f, ast.FunctionDef) and f.name == "update"][0]
    exprs = [st for st in init.body if isinstance(st, ast.Assign) and isinstance(st.targets[0], ast.Attribute) and st.targets[0].attr in ("mean", "std")]
    assert [e.targets[0].attr for e in exprs] == ["mean", "std"]
    mean_std = compile(ast.Module(body=exprs, type_ignores=[]), "misc_util.py", "exec")
    upd_code = compile(ast.Module(body=[upd], type_ignores=[]), "misc_util.py", "exec")
    ns = {"np": np, "MPI": types.SimpleNamespace(COMM_WORLD=OneRank(), SUM=None), "tf": EagerTf}
    exec(upd_code, ns)

    class Rms(object):                                           # the three TF variables as numpy, initialised as src/utils/misc_util.py:36-50 does
        def __init__(self, shape, epsilon=1e-2):
            self.shape = shape
            self._sum = np.zeros(shape, np.float64); self._sumsq = np.full(shape, epsilon, np.float64); self._count = np.float64(epsilon)

        def incfiltparams(self, s, ss, c):
            self._sum = self._sum + s; self._sumsq = self._sumsq + ss; self._count = self._count + c
        update = ns["update"]

    r = Rms((56,))
    batches = [rng.randn(17, 56) * 3 + 1, rng.randn(400, 56) * np.linspace(0.001, 5, 56), rng.randn(1, 56)]
    states = []
    exec(mean_std, {"tf": EagerTf, "self": r}); states.append((r.mean.copy(), r.std.copy()))
    for xb in batches:
        r.update(xb.astype(np.float32))
        exec(mean_std, {"tf": EagerTf, "self": r}); states.append((r.mean.copy(), r.std.copy()))
    for k, xb in enumerate(batches):
        out["rms_x%d" % k] = xb.astype(np.float32)
    out["rms_mean"] = np.stack([s[0] for s in states]); out["rms_std"] = np.stack([s[1] for s in states])
    out["rms_sum"], out["rms_sumsq"], out["rms_count"] = r._sum, r._sumsq, np.array(r._count)
    r2 = Rms((4,))                                                # nearly constant data: the variance floor (std = sqrt(1e-2))
    xc = (np.array([0.3, -2.0, 5.0, 0.0]) + 1e-3 * rng.randn(300, 4)).astype(np.float32)
    r2.update(xc)
    exec(mean_std, {"tf": EagerTf, "self": r2})
    out["rms_floor_x"], out["rms_floor_mean"], out["rms_floor_std"] = xc, r2.mean.copy(), r2.std.copy()

    np.savez_compressed(OUT, **out)
    print("wrote %s: %d arrays, %.1f KB" % (OUT, len(out), os.path.getsize(OUT) / 1e3))


if __name__ == "__main__":
    main()
