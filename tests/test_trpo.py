"""TRPO learner: numerics of its pieces against plain restatements of src/trpo.py, src/cg.py,
src/mpi_adam.py, src/distributions.py, and an end-to-end improvement test on a scripted environment."""
import math

import numpy as np
import pytest
import torch

from deepmimic_mujoco_amd.policy import MlpPolicy
from deepmimic_mujoco_amd.trpo import TrpoLearner, MpiAdam, cg, flat, learn, explained_variance, POL_KEYS


def test_cg_matches_reference_iteration_and_solves_spd():
    rng = np.random.RandomState(0)
    B = rng.randn(12, 12); A = B @ B.T + 12 * np.eye(12); b = rng.randn(12)

    def cg_np(f_Ax, b, cg_iters=10, residual_tol=1e-10):       # src/cg.py:2-34 restated
        p = b.copy(); r = b.copy(); x = np.zeros_like(b); rdotr = r.dot(r)
        for _ in range(cg_iters):
            z = f_Ax(p); v = rdotr / p.dot(z); x += v * p; r -= v * z
            newrdotr = r.dot(r); mu = newrdotr / rdotr; p = r + mu * p; rdotr = newrdotr
            if rdotr < residual_tol:
                break
        return x
    At = torch.from_numpy(A)
    for iters in (3, 10, 40):
        x = cg(lambda p: At @ p, torch.from_numpy(b), cg_iters=iters)
        assert np.allclose(x.numpy(), cg_np(lambda p: A @ p, b, iters), rtol=1e-9, atol=1e-12)
    assert np.allclose(cg(lambda p: At @ p, torch.from_numpy(b), cg_iters=40).numpy(), np.linalg.solve(A, b), atol=1e-5)   # stops at r.r < 1e-10, as the reference
    # the GPU form applies the residual test without reading it back: same x as the break, at every tolerance (incl. one that stops after a single iteration)
    for tol in (1e-10, 1e-3, 1e3):
        x_break = cg(lambda p: At @ p, torch.from_numpy(b), cg_iters=40, residual_tol=tol, sync_free=False)
        x_mask = cg(lambda p: At @ p, torch.from_numpy(b), cg_iters=40, residual_tol=tol, sync_free=True)
        assert torch.isfinite(x_mask).all() and torch.equal(x_break, x_mask)


def test_distribution_formulas_match_reference_definitions():
    rng = np.random.RandomState(1)
    m0, m1 = rng.randn(5, 28), rng.randn(5, 28)
    l0, l1 = rng.randn(1, 28) * 0.3, rng.randn(1, 28) * 0.3
    x = rng.randn(5, 28)
    kl = TrpoLearner._kl(*(torch.from_numpy(a) for a in (m0, l0, m1, l1))).numpy()
    ref = (l1 - l0 + (np.exp(l0) ** 2 + (m0 - m1) ** 2) / (2 * np.exp(l1) ** 2) - 0.5).sum(-1)     # distributions.py:235-237
    assert np.allclose(kl, ref, rtol=1e-12)
    nl = TrpoLearner._neglogp(torch.from_numpy(x), torch.from_numpy(m0), torch.from_numpy(l0)).numpy()
    ref = 0.5 * (((x - m0) / np.exp(l0)) ** 2).sum(-1) + 0.5 * np.log(2 * np.pi) * 28 + l0.sum(-1)  # :231-234
    assert np.allclose(nl, ref, rtol=1e-12)
    assert np.all(TrpoLearner._kl(*(torch.from_numpy(a) for a in (m0, l0, m0, l0))).numpy() == 0)


def _segment(pi, n=48, T=16, seed=0):
    g = torch.Generator(); g.manual_seed(seed)
    ob = torch.randn((T, n, 56), generator=g)
    pi.seed(seed)
    with torch.no_grad():
        ac, vpred = pi.act(True, ob.reshape(-1, 56))
    return {"ob": ob, "ac": ac.to(torch.float32).reshape(T, n, 28), "vpred": vpred.reshape(T, n),
            "rew": torch.rand((T, n), generator=g), "new": (torch.rand((T, n), generator=g) < 0.1).to(torch.int32),
            "nextvpred": torch.zeros(n)}


def test_fisher_vector_product_is_the_kl_hessian():
    pi = MlpPolicy(seed=3)
    L = TrpoLearner(pi, cg_damping=0.0)
    ob = torch.randn(64, 56)
    with torch.no_grad():
        old_mean, _ = pi.forward(ob); old_logstd = pi.params["logstd"].clone()
    mean, logstd = L._pd(ob)
    kl = L._kl(old_mean, old_logstd, mean, logstd).mean()
    klgrads = flat(torch.autograd.grad(kl, L.pol, create_graph=True))
    v = torch.randn_like(klgrads); v = v / v.norm()
    hv = flat(torch.autograd.grad(klgrads.dot(v), L.pol, retain_graph=True))
    # finite-difference Hessian-vector product of the same KL
    th = L.get_flat()

    def grad_at(theta):
        L.set_from_flat(theta)
        m, ls = L._pd(ob)
        return flat(torch.autograd.grad(L._kl(old_mean, old_logstd, m, ls).mean(), L.pol)).to(torch.float64)
    eps = 5e-2
    fd = (grad_at(th + eps * v) - grad_at(th - eps * v)) / (2 * eps)
    L.set_from_flat(th)
    assert float((fd - hv.to(torch.float64)).norm() / hv.norm()) < 2e-2
    assert float(v.dot(hv)) > 0                                  # Fisher matrix: positive (semi-)definite


def test_mpi_adam_update_rule():
    p = torch.tensor([1.0, -2.0, 0.5], requires_grad=True)
    opt = MpiAdam([p])
    th = p.detach().numpy().astype(np.float32).copy(); m = np.zeros(3, np.float32); v = np.zeros(3, np.float32)
    rng = np.random.RandomState(0)
    for t in range(1, 6):
        g = rng.randn(3).astype(np.float32)
        opt.update(torch.from_numpy(g), 1e-3)
        a = 1e-3 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)      # src/mpi_adam.py:30-34
        m = 0.9 * m + 0.1 * g; v = 0.999 * v + 0.001 * g * g
        th = th + (-a) * m / (np.sqrt(v) + 1e-8)
        assert np.allclose(p.detach().numpy(), th, rtol=1e-5, atol=1e-7)


def test_update_respects_the_trust_region_and_improves_the_surrogate():
    pi = MlpPolicy(seed=5)
    L = TrpoLearner(pi, vf_batch_size=128)
    seg = _segment(pi)
    seg["rew"] = -((seg["ac"] - 0.3) ** 2).mean(-1)             # reward prefers actions near 0.3
    before = L.get_flat()
    st = L.update(seg)
    assert st["stepsize"] > 0 and 0 < st["meankl"] <= 1.5 * 0.01 + 1e-6 and st["improve"] >= 0
    assert not torch.equal(before, L.get_flat())
    assert abs(st["entropy"] - pi.entropy()) < 0.5 and math.isfinite(st["ev_tdlam_before"])
    assert float(pi.ob_rms.count) > 48 * 16                       # obs filter saw the batch (+ the vf minibatches)


class _ToyVecEnv:
    """obs ~ N(0,1) i.i.d.; reward = -|a - 0.5 obs[:28]|^2 / 28; never done.  A policy can only score by reading the obs."""

    def __init__(self, n, seed=0):
        self.num_envs = n
        self.g = torch.Generator(); self.g.manual_seed(seed)
        self.batch = self
        self.ob = None

    def reset(self, mode, out=None):
        self.ob = torch.randn((self.num_envs, 56), generator=self.g, dtype=torch.float64).numpy()
        out[...] = self.ob
        return out

    def step(self, ac, nsub, out):
        ob, rew, done = out
        rew[...] = -((np.asarray(ac) - 0.5 * self.ob[:, :28]) ** 2).mean(-1)
        self.ob = torch.randn((self.num_envs, 56), generator=self.g, dtype=torch.float64).numpy()
        ob[...] = self.ob; done[...] = 0
        return out


def test_learn_improves_return_on_a_scripted_env(tmp_path):
    torch.manual_seed(0)
    pi = MlpPolicy(seed=7); pi.seed(7)
    hist = learn(_ToyVecEnv(256, 1), pi, timesteps_per_batch=16, max_iters=12, log=None, gamma=0.0, lam=0.0, vf_batch_size=1024,
                 log_dir=str(tmp_path))
    from deepmimic_mujoco_amd.logio import read_progress_csv, read_monitor_csv
    kv = read_progress_csv(str(tmp_path / "progress.csv"))
    assert list(kv)[:3] == ["EpRewMean", "EpThisIter", "TimestepsSoFar"] and len(kv["meankl"]) == 12 and kv["TimestepsSoFar"][-1] == 12 * 16 * 256
    hdr, r, l, t = read_monitor_csv(str(tmp_path / "monitor.json.monitor.csv"))
    assert "t_start" in hdr and len(r) == 0            # the toy env never ends an episode
    assert len(hist) == 12 and hist[-1]["TimestepsSoFar"] == 12 * 16 * 256
    # with gamma = 0 the advantage is the immediate reward: the surrogate keeps finding improvement and the policy mean moves
    ob = torch.randn(4096, 56)
    with torch.no_grad():
        ac, _ = pi.act(False, ob)
    err_after = float(((ac - 0.5 * ob[:, :28].to(torch.float64)) ** 2).mean())
    err_init = float((0.5 * ob[:, :28] ** 2).mean()) * 0.5          # policy mean ~ 0 at init: E|0.5 ob|^2 = 0.25
    assert err_after < 0.8 * 0.25, (err_after, err_init)
    assert all(h["meankl"] <= 0.0151 for h in hist)


def test_runner_counts_each_envs_first_episode():
    """`runner` (src/trpo.py:356-436 for a batch): every env contributes its FIRST episode — steps and rewards stop counting at its
    first `done` although the env goes on — and the cap is `timesteps_per_batch + 1` steps."""
    from deepmimic_mujoco_amd.trpo import runner

    class Scripted:
        num_envs = 4

        def __init__(self):
            self.t = 0

        def reset(self, mode, out=None):
            assert mode == "init"
            self.t = 0
            out[...] = 0.0
            return out

        def step(self, ac):
            self.t += 1
            done = np.array([self.t == 3, self.t == 5, self.t in (2, 4), False])         # env 2 "ends" twice: only the first counts
            return np.full((4, 56), float(self.t)), np.array([1.0, 2.0, 0.5, 1.0]), done, [{}] * 4

    pi = MlpPolicy(seed=0); pi.seed(0)
    lines = []
    avg_len, avg_ret, lens, rets = runner(Scripted(), pi, timesteps_per_batch=9, stochastic_policy=True, log=lines.append)
    assert list(lens) == [3, 5, 2, 10] and list(rets) == [3.0, 10.0, 1.0, 10.0]
    assert avg_len == 5.0 and avg_ret == 6.0 and lines[0] == "stochastic policy:" and lines[1] == "Average length: 5.0"


def test_explained_variance():
    y = torch.tensor([1.0, 2.0, 3.0, 4.0])
    assert explained_variance(y, y) == 1.0 and abs(explained_variance(torch.zeros(4), y)) < 1e-12


# ---- numeric pin of one update against the analytic float64 restatement (tests/trpo_numpy.py, fixture trpo_update_golden.npz) ----
def _golden_update(device, **learner_kw):
    """Run TrpoLearner.update on the fixture's segment; return (learner, policy, stats, fixture)."""
    import os
    from deepmimic_mujoco_amd.trpo import TrpoLearner, POL_KEYS, VF_KEYS
    from deepmimic_mujoco_amd.policy import MlpPolicy
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trpo_update_golden.npz"))
    pi = MlpPolicy(device=device, seed=0)
    with torch.no_grad():
        for k in POL_KEYS + VF_KEYS:
            pi.params[k].copy_(torch.as_tensor(g["p0/" + k], dtype=torch.float32).reshape(pi.params[k].shape))
    pi.ob_rms.sum = torch.as_tensor(g["rms0_sum"], dtype=torch.float64, device=device)
    pi.ob_rms.sumsq = torch.as_tensor(g["rms0_sumsq"], dtype=torch.float64, device=device)
    pi.ob_rms.count = torch.as_tensor(float(g["rms0_count"]), dtype=torch.float64, device=device)
    pi.ob_rms._refresh()
    learner = TrpoLearner(pi, vf_batch_size=128, **learner_kw)
    perms = [torch.as_tensor(p, dtype=torch.int64) for p in g["perms"]]
    learner.perm_source = lambda n, it=iter(perms): next(it)
    t = lambda a, dt: torch.as_tensor(a, dtype=dt, device=device)
    seg = {"ob": t(g["ob"], torch.float32), "ac": t(g["ac"], torch.float32), "rew": t(g["rew"], torch.float32), "vpred": t(g["vpred"], torch.float32),
           "new": t(g["new"], torch.int32), "nextvpred": t(g["nextvpred"], torch.float32), "ep_lens": [], "ep_rets": []}
    stats = learner.update(seg)
    return learner, pi, stats, g, seg


def _check_against_golden(learner, pi, stats, g, seg):
    from deepmimic_mujoco_amd.trpo import POL_KEYS, VF_KEYS, flat
    from tests import trpo_numpy as TN
    T, N = int(g["T"]), int(g["N"])
    # GAE of the segment (src/trpo.py:83-94)
    assert np.allclose(seg["adv"].cpu().numpy(), g["adv"], rtol=2e-5, atol=2e-4) and np.allclose(seg["tdlamret"].cpu().numpy(), g["tdlamret"], rtol=2e-5, atol=2e-4)
    rel = lambda a, b: float(np.linalg.norm(np.asarray(a, dtype=np.float64) - b) / np.linalg.norm(b))
    cos = lambda a, b: float(np.dot(np.asarray(a, dtype=np.float64), b) / (np.linalg.norm(a) * np.linalg.norm(b)))
    L = learner.last
    # surrogate gradient, CG step direction (10 iterations amplify float32 rounding), scaled step
    assert rel(L["g"].cpu().numpy(), g["st/g"]) < 2e-4, rel(L["g"].cpu().numpy(), g["st/g"])
    assert cos(L["stepdir"].cpu().numpy(), g["st/stepdir"]) > 0.9999 and rel(L["stepdir"].cpu().numpy(), g["st/stepdir"]) < 2e-2
    assert abs(L["shs"] / float(g["st/shs"]) - 1) < 2e-2 and abs(L["lm"] / float(g["st/lm"]) - 1) < 1e-2
    assert cos(L["fullstep"].cpu().numpy(), g["st/fullstep"]) > 0.9999
    # line search: same number of halvings, same KL / surrogate at the accepted point
    assert stats["stepsize"] == float(g["st/stepsize"]) == 0.5
    assert abs(stats["expectedimprove"] / float(g["st/expectedimprove"]) - 1) < 1e-2
    assert abs(stats["meankl"] - float(g["st/kl"])) < 2e-4 and abs(stats["surrgain"] - float(g["st/surr"])) < 2e-3
    assert stats["meankl"] <= 0.015
    # parameters after the update: policy (theta + stepsize * fullstep) and value function (3 epochs x 4 Adam steps)
    p1 = {k: g["p1/" + k] for k in POL_KEYS + VF_KEYS}
    th = flat([pi.params[k].detach() for k in POL_KEYS]).cpu().numpy()
    assert rel(th - TN.flat({k: g["p0/" + k] for k in POL_KEYS}, POL_KEYS), TN.flat(p1, POL_KEYS) - TN.flat({k: g["p0/" + k] for k in POL_KEYS}, POL_KEYS)) < 2e-2
    vf = flat([pi.params[k].detach() for k in VF_KEYS]).cpu().numpy()
    dv_ref = TN.flat(p1, VF_KEYS) - TN.flat({k: g["p0/" + k] for k in VF_KEYS}, VF_KEYS)
    assert rel(vf - TN.flat({k: g["p0/" + k] for k in VF_KEYS}, VF_KEYS), dv_ref) < 2e-2 and np.abs(dv_ref).max() > 5e-3
    # the observation filter saw the batch once and every value-fit minibatch once (src/trpo.py:242,293)
    assert abs(float(pi.ob_rms.count) - float(g["rms1_count"])) < 1e-6 and float(g["rms1_count"]) - float(g["rms0_count"]) == 4 * T * N
    assert np.allclose(pi.ob_rms.sum.cpu().numpy(), g["rms1_sum"], rtol=1e-9, atol=1e-6)


def test_numpy_restatement_reproduces_its_committed_fixture():
    """tests/golden/trpo_update_golden.npz was written by tests/golden/gen/make_trpo_fixture.py from tests/trpo_numpy.py: re-running
    the analytic float64 update on the fixture's inputs gives the stored outputs (the fixture is not stale), and its pieces satisfy
    their definitions: F is symmetric positive, CG reduces the residual, the accepted step respects the trust region."""
    import os
    from tests import trpo_numpy as TN
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trpo_update_golden.npz"))
    T, N = int(g["T"]), int(g["N"])
    p0 = {k[3:]: g[k] for k in g.files if k.startswith("p0/")}
    rms = TN.Rms(); rms.sum = g["rms0_sum"].copy(); rms.sumsq = g["rms0_sumsq"].copy(); rms.count = float(g["rms0_count"])
    fl = lambda a: np.swapaxes(a, 0, 1).reshape((T * N,) + a.shape[2:])
    p1, st = TN.update(p0, rms, fl(g["ob"]), fl(g["ac"]), fl(g["adv"]), fl(g["tdlamret"]), list(g["perms"]), vf_batch=128)
    for k in ("g", "stepdir", "fullstep"):
        assert np.array_equal(st[k], g["st/" + k])
    assert st["stepsize"] == float(g["st/stepsize"]) and st["kl"] == float(g["st/kl"]) and all(np.array_equal(p1[k], g["p1/" + k]) for k in p1)
    assert st["kl"] <= 0.015 and st["surr"] >= st["surrbefore"] and st["expectedimprove"] > 0


def test_learner_update_matches_the_float64_restatement_on_cpu():
    _check_against_golden(*_golden_update("cpu"))


@pytest.mark.gpu
def test_learner_update_matches_the_float64_restatement_on_gpu():
    """src/trpo.py:235-296 on torch-ROCm: gradient, CG direction, step scaling, line-search outcome, KL and value-fit parameters
    against the analytic float64 restatement (committed fixture), float32 tolerances."""
    _check_against_golden(*_golden_update("cuda:0"))


@pytest.mark.gpu
def test_native_value_fit_equals_the_eager_loop_on_gpu():
    """csrc/vf_kernel.h (dm_vf_fit_epoch: obs filter, forward / backward of the value net, MpiAdam, three launches per minibatch)
    against the eager torch loop on the same update (3 epochs x 4 minibatches of 128): same value parameters, Adam moments and
    obs-filter state to float32 rounding; then a large minibatch (4 096 samples, 128 blocks of partial gradients) on random data."""
    from deepmimic_mujoco_amd.trpo import VF_KEYS, flat, TrpoLearner
    from deepmimic_mujoco_amd.policy import MlpPolicy
    le, pe, _, g, _ = _golden_update("cuda:0", vf_graph=False, vf_native=False)
    ln, pn, _, _, _ = _golden_update("cuda:0", vf_native=True)
    assert ln._vf_scratch is not None and le._vf_scratch is None
    ve = flat([pe.params[k].detach() for k in VF_KEYS]); vn = flat([pn.params[k].detach() for k in VF_KEYS])
    dv = ve - flat([torch.as_tensor(g["p0/" + k], dtype=torch.float32, device="cuda:0").reshape(-1) for k in VF_KEYS])
    assert float((ve - vn).abs().max()) < 2e-3 * float(dv.abs().max()) and float(dv.abs().max()) > 5e-3      # the 12 Adam steps agree to 0.2 %
    assert torch.allclose(pe.ob_rms.sum, pn.ob_rms.sum, rtol=1e-12, atol=1e-8) and float(pe.ob_rms.count) == float(pn.ob_rms.count)
    assert torch.allclose(pe.ob_rms.std, pn.ob_rms.std, rtol=1e-6) and le.vfadam.t == ln.vfadam.t == 12
    assert float((le.vfadam.m - ln.vfadam.m).abs().max()) < 1e-5 * max(1e-3, float(le.vfadam.m.abs().max())) + 1e-7
    # one big minibatch: gradient (through the first Adam step: m = 0.1 g) against autograd
    torch.manual_seed(0)
    n = 4096
    ob = torch.randn(n, 56, device="cuda:0") * 2.0; ret = torch.randn(n, device="cuda:0") * 3.0
    outs = []
    for native in (False, True):
        pi = MlpPolicy(device="cuda:0", seed=3)
        L = TrpoLearner(pi, vf_batch_size=n, vf_iters=1, vf_graph=False, vf_native=native)
        L.perm_source = lambda k: torch.arange(k)
        if native:
            assert L._vf_native_ready(ob, ret)
            L._vf_native_epoch(ob, ret, torch.arange(n, device="cuda:0"), n)
        else:
            L._vf_step(ob, ret)
        outs.append((L.vfadam.m.clone(), flat([pi.params[k].detach() for k in VF_KEYS]), pi.ob_rms.mean.clone(), pi.ob_rms.std.clone()))
    (m0, t0, mu0, sd0), (m1, t1, mu1, sd1) = outs
    assert float((m0 - m1).abs().max()) < 2e-5 * float(m0.abs().max()) + 1e-8, float((m0 - m1).abs().max()) / float(m0.abs().max())
    assert float((t0 - t1).abs().max()) < 1e-5 and torch.allclose(mu0, mu1, atol=1e-6) and torch.allclose(sd0, sd1, rtol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("bs", [1000, 37])
def test_native_value_fit_handles_ragged_minibatches_on_gpu(bs):
    """Minibatch sizes that are not a multiple of the 32-sample blocks (a partial last block, partial row groups of the filter sums):
    two epochs of three minibatches against the eager loop."""
    from deepmimic_mujoco_amd.trpo import VF_KEYS, flat, TrpoLearner
    from deepmimic_mujoco_amd.policy import MlpPolicy
    torch.manual_seed(1)
    n = 3 * bs + 5                                               # the last 5 samples are dropped (include_final_partial_batch=False)
    ob = torch.randn(n, 56, device="cuda:0") * 1.5 + 0.3; ret = torch.randn(n, device="cuda:0") * 2.0
    outs = []
    for native in (False, True):
        pi = MlpPolicy(device="cuda:0", seed=5)
        L = TrpoLearner(pi, vf_batch_size=bs, vf_iters=2, vf_graph=False, vf_native=native)
        perms = [torch.randperm(n, generator=torch.Generator().manual_seed(k)) for k in (7, 8)]
        for pm in perms:
            inds = pm.to("cuda:0")
            if native:
                L._vf_native_epoch(ob, ret, inds, bs)
            else:
                for o in range(0, n - bs + 1, bs):
                    mb = inds[o:o + bs]
                    L._vf_step(ob[mb], ret[mb])
        outs.append((flat([pi.params[k].detach() for k in VF_KEYS]), L.vfadam.m.clone(), pi.ob_rms.sum.clone(), float(pi.ob_rms.count), L.vfadam.t))
    (t0, m0, s0, c0, k0), (t1, m1, s1, c1, k1) = outs
    assert k0 == k1 == 6 and c0 == c1 and torch.allclose(s0, s1, rtol=1e-12, atol=1e-9)
    assert float((m0 - m1).abs().max()) < 1e-4 * float(m0.abs().max()) + 1e-8 and float((t0 - t1).abs().max()) < 2e-5


@pytest.mark.gpu
def test_value_fit_as_captured_graph_equals_the_eager_loop_on_gpu():
    """The value fit's minibatch step replayed as a captured hipGraph (single-process GPU runs) against the eager loop on the same
    update: same value parameters, same obs-filter moments, same Adam state; and the graph path really ran."""
    from deepmimic_mujoco_amd.trpo import VF_KEYS, flat
    le, pe, se, g, _ = _golden_update("cuda:0", vf_graph=False, vf_native=False)
    lg, pg, sg, _, _ = _golden_update("cuda:0", vf_graph=True, vf_native=False)
    assert le._vfg is None and lg._vfg is not None and lg._vfg.ok
    ve = flat([pe.params[k].detach() for k in VF_KEYS]); vg = flat([pg.params[k].detach() for k in VF_KEYS])
    assert float((ve - vg).abs().max()) < 1e-6 * max(1.0, float(ve.abs().max()))
    assert torch.allclose(pe.ob_rms.sum, pg.ob_rms.sum, rtol=1e-12, atol=1e-9) and float(pe.ob_rms.count) == float(pg.ob_rms.count)
    assert le.vfadam.t == lg.vfadam.t == 12 and float((le.vfadam.m - lg.vfadam.m).abs().max()) < 1e-7


# ---- the policy half as hand-written kernels (csrc/pg_kernel.h) -------------------------------------------------------------------
def _pg_case(n, seed=0):
    """Random batch + policy for the native policy-gradient kernels, with the float64 numpy restatement's view of the same numbers."""
    from deepmimic_mujoco_amd.trpo import TrpoLearner, POL_KEYS
    from deepmimic_mujoco_amd.policy import MlpPolicy
    from tests import trpo_numpy as TN
    rng = np.random.RandomState(seed)
    pi = MlpPolicy(device="cuda:0", seed=seed)
    with torch.no_grad():
        pi.params["logstd"].copy_(torch.as_tensor(rng.uniform(-0.7, 0.3, (1, 28)), dtype=torch.float32))
        pi.params["polfinal/w"].mul_(20.0)                                  # (the initial 0.01 scale would make every mean ~0)
        for k in ("polfc1/b", "polfc2/b", "polfinal/b"):
            pi.params[k].copy_(torch.as_tensor(0.1 * rng.randn(*pi.params[k].shape), dtype=torch.float32))
    ob = (rng.randn(n, 56) * np.linspace(0.5, 3.0, 56) + rng.randn(56)).astype(np.float32)
    pi.ob_rms.update(torch.as_tensor(ob, device="cuda:0"))
    ac = rng.randn(n, 28).astype(np.float32)
    atarg = rng.randn(n).astype(np.float32)
    p = {k: pi.params[k].detach().cpu().numpy().astype(np.float64) for k in POL_KEYS}
    rms = TN.Rms(); rms.sum = pi.ob_rms.sum.cpu().numpy().copy(); rms.sumsq = pi.ob_rms.sumsq.cpu().numpy().copy(); rms.count = float(pi.ob_rms.count)
    learner = TrpoLearner(pi, pg_native=True)
    return learner, pi, p, rms, ob, ac, atarg


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 33, 1000, 32 * 256 * 3 + 7])    # one sample; one tile + one row; a ragged last tile; more tiles than blocks (every block loops) + a ragged tail
def test_native_policy_gradient_and_fisher_product_match_the_analytic_float64_formulas_on_gpu(n):
    """dm_pg_losses / dm_pg_fvp against tests/trpo_numpy.py (src/trpo.py:224-230 written out by hand in float64): surrogate gradient at
    pi == oldpi, F v = J^T Sigma^-1 J v / N_f (+) 2 v on every 5th sample, and the losses at a moved policy — float32 tolerances."""
    from deepmimic_mujoco_amd.trpo import POL_KEYS
    from tests import trpo_numpy as TN
    learner, pi, p, rms, ob, ac, atarg = _pg_case(n)
    dev = "cuda:0"
    t = lambda a: torch.as_tensor(a, device=dev)
    ob_t, ac_t, at_t = t(ob), t(ac), t(atarg)
    theta0 = learner.get_flat().contiguous()
    old_logstd = pi.params["logstd"].detach().reshape(-1).clone()
    old_mean = torch.empty((n, 28), dtype=torch.float32, device=dev)
    losses, g = learner._pg_losses(ob_t, ac_t, at_t, old_mean, old_logstd, theta0, write_old=True, with_grad=True)
    # float64 side
    x = TN.obz(ob.astype(np.float64), rms)
    m0, cache = TN.pol_forward(p, x)
    sig2 = np.exp(2 * p["logstd"])
    a64, ac64 = atarg.astype(np.float64), ac.astype(np.float64)
    dmean = a64[:, None] * (ac64 - m0) / sig2 / n
    dls = (a64[:, None] * ((ac64 - m0) ** 2 / sig2 - 1.0)).sum(0) / n
    g_ref = TN.pol_backward(p, cache, dmean, dls)
    rel = lambda a, b: float(np.linalg.norm(np.asarray(a, dtype=np.float64) - b) / np.linalg.norm(b))
    assert rel(old_mean.cpu().numpy(), m0) < 1e-5
    assert rel(g.cpu().numpy(), g_ref) < 2e-4, rel(g.cpu().numpy(), g_ref)
    L = losses.cpu().numpy()
    assert abs(L[3] - a64.mean()) < 1e-5 and abs(L[1]) < 1e-6 and abs(L[0] - L[3] - L[2]) < 1e-6      # ratio == 1, KL == 0 at pi == oldpi
    # Fisher-vector product on every 5th sample
    rng = np.random.RandomState(3)
    v = rng.randn(theta0.numel()) * np.abs(g_ref).mean()
    xs = x[::5]; _, cache_f = TN.pol_forward(p, xs)
    vd = TN.unflat(v, p, POL_KEYS)
    fv_ref = TN.pol_backward(p, cache_f, TN.pol_jvp(p, cache_f, vd) / sig2 / xs.shape[0], 2.0 * vd["logstd"].reshape(-1))
    fv = learner._pg_fvp(ob_t, theta0, t(v.astype(np.float32)))
    assert rel(fv.cpu().numpy(), fv_ref) < 2e-4, rel(fv.cpu().numpy(), fv_ref)
    v_t = t(v.astype(np.float32))
    assert float(v_t.dot(fv)) > 0                                             # positive definite along v
    # ... and against double back-propagation through torch's graph (the path it replaces)
    learner2_fv = None
    with torch.enable_grad():
        mean_f, logstd_f = learner._pd(ob_t[::5])
        kl_f = learner._kl(old_mean[::5], pi.params["logstd"].detach(), mean_f, logstd_f).mean()
        klg = torch.cat([q.reshape(-1) for q in torch.autograd.grad(kl_f, learner.pol, create_graph=True)])
        learner2_fv = torch.cat([q.reshape(-1) for q in torch.autograd.grad(klg.dot(v_t), learner.pol)])
    assert rel(fv.cpu().numpy(), learner2_fv.double().cpu().numpy()) < 5e-4
    # losses at a moved policy (the line search's evaluation, :262-283)
    step = 0.05 * rng.randn(theta0.numel()) / np.sqrt(theta0.numel())
    q = dict(p); q.update(TN.unflat(TN.flat(p, POL_KEYS) + step, p, POL_KEYS))
    m1, _ = TN.pol_forward(q, x)
    ratio = np.exp(TN.neglogp(ac64, m0, p["logstd"]) - TN.neglogp(ac64, m1, q["logstd"]))
    surr_ref, kl_ref = float((ratio * a64).mean()), float(TN.kl(m0, p["logstd"], m1, q["logstd"]).mean())
    theta1 = (theta0.double() + t(step)).float().contiguous()
    l1, _ = learner._pg_losses(ob_t, ac_t, at_t, old_mean, old_logstd, theta1, write_old=False, with_grad=False)
    l1 = l1.cpu().numpy()
    assert abs(l1[3] - surr_ref) < 2e-4 * max(1.0, abs(surr_ref)) and abs(l1[1] - kl_ref) < 2e-4 * max(1e-2, kl_ref), (l1, surr_ref, kl_ref)
    # deterministic: the same launch twice gives the same bits
    _, g2 = learner._pg_losses(ob_t, ac_t, at_t, old_mean, old_logstd, theta0, write_old=False, with_grad=True)
    assert torch.equal(g, g2) and torch.equal(fv, learner._pg_fvp(ob_t, theta0, v_t))


@pytest.mark.gpu
@pytest.mark.parametrize("blocks", [1, 7, 128])
def test_native_policy_kernels_on_a_capped_grid_on_gpu(blocks):
    """dm_pg_losses / dm_pg_fvp with max_blocks (the CUs the learner leaves to the value fit beside it): the same sums in another block
    order — equal to the full grid's within float32 rounding, bit-identical launch to launch, and the losses (float64 sums) almost exactly."""
    n = 32 * 300 + 5
    learner, pi, p, rms, ob, ac, atarg = _pg_case(n, seed=4)
    dev = "cuda:0"
    t = lambda a: torch.as_tensor(a, device=dev)
    ob_t, ac_t, at_t = t(ob), t(ac), t(atarg)
    theta0 = learner.get_flat().contiguous()
    old_logstd = pi.params["logstd"].detach().reshape(-1).clone()
    old_mean = torch.empty((n, 28), dtype=torch.float32, device=dev)
    v = torch.randn(theta0.numel(), device=dev, generator=torch.Generator(device=dev).manual_seed(1)) * 1e-2

    def run(cap):
        learner._share = None if cap is None else {"t": 0.0, "until": 1e9, "blocks": cap}
        om = torch.empty_like(old_mean)
        losses, g = learner._pg_losses(ob_t, ac_t, at_t, om, old_logstd, theta0, write_old=True, with_grad=True)
        fv = learner._pg_fvp(ob_t, theta0, v)
        learner._share = None
        return losses, g, fv, om
    full, capped, again = run(None), run(blocks), run(blocks)
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    assert rel(capped[1], full[1]) < 2e-5 and rel(capped[2], full[2]) < 2e-5
    assert torch.equal(capped[3], full[3])                                      # the means do not depend on the grid
    assert torch.allclose(capped[0], full[0], rtol=1e-6, atol=1e-7)
    for a, b in zip(capped, again):
        assert torch.equal(a, b)


def test_shuffles_drawn_ahead_are_the_ones_an_update_would_draw_itself():
    """TrpoLearner._prefetch_perms / _next_perm: the value fit's shuffles come from one generator in one order whether they are drawn ahead
    (after the previous update) or on demand; a changed segment size drops what was drawn ahead and goes on from the generator's state."""
    n = 1000
    def learner():
        return TrpoLearner(MlpPolicy(device="cpu", seed=0), vf_iters=3, seed=7)
    a, b = learner(), learner()
    want = [b._next_perm(n, torch.device("cpu")) for _ in range(6)]                       # on demand
    a._prefetch_perms(n, torch.device("cpu"))
    assert len(a._perms) == 3
    got = [a._next_perm(n, torch.device("cpu")) for _ in range(3)]
    a._prefetch_perms(n, torch.device("cpu")); a._prefetch_perms(n, torch.device("cpu"))  # (a second call with a full list draws nothing)
    got += [a._next_perm(n, torch.device("cpu")) for _ in range(3)]
    assert all(torch.equal(x, y) for x, y in zip(got, want)) and a._perms == []
    a._prefetch_perms(n, torch.device("cpu"))
    p = a._next_perm(n + 1, torch.device("cpu"))                                          # another size: the three drawn ahead are dropped
    assert p.numel() == n + 1 and sorted(p.tolist()) == list(range(n + 1)) and a._perms == []
    a.perm_source = lambda k: torch.arange(k)
    a._prefetch_perms(n, torch.device("cpu"))                                             # tests' injected shuffles: nothing is drawn ahead
    assert a._perms == []


def test_cu_sharing_plan_is_a_function_of_the_launch_sequence():
    """TrpoLearner._pg_share_begin / _pg_grid: which policy launches run on the narrow grid beside the value fit comes from a cost model of
    the launches issued so far — the same sequence gives the same grids (a seeded run reproduces), the narrow phase ends once the modelled
    fit is over, and a fit that wants the whole chip turns sharing off."""
    pi = MlpPolicy(device="cpu", seed=0)
    L = TrpoLearner(pi, vf_batch_size=4096, vf_iters=1)                      # (one epoch: the modelled fit ends inside the CG iterations)
    n = 4096 * 128

    def plan():
        L._pg_share_begin(n, 4096)
        seq = [L._pg_grid(L.PG_GRAD_NS * n)] + [L._pg_grid(L.PG_FVP_NS * (n // 5)) for _ in range(11)] + [L._pg_grid(L.PG_LOSS_NS * n) for _ in range(3)]
        L._share = None
        return seq
    a, b = plan(), plan()
    assert a == b and a[0] == 128 and a[-1] == 0                              # narrow first (the fit runs), every CU at the end
    k = a.index(0)
    assert all(x == 128 for x in a[:k]) and all(x == 0 for x in a[k:])        # one switch, never back
    L._pg_share_begin(n, 8192)                                                # 256 blocks of the fit's gradient kernel: nothing to leave
    assert L._share is None and L._pg_grid(1e6) == 0
    L._pg_share_begin(1024, 128)                                              # the reference's sizes: four blocks for the fit, 252 for the policy step
    assert L._share["blocks"] == 252


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 255, 4096 * 128 + 3])
def test_native_obs_filter_update_equals_running_mean_std_on_gpu(n):
    """dm_rms_update (TrpoLearner._rms_update) against RunningMeanStd.update (src/utils/misc_util.py:47-70): float64 sums of a float32 batch
    (another summation order: equal to ~1e-13 relative), the count exactly, mean / std to float32 rounding — twice in a row (state carried)."""
    from deepmimic_mujoco_amd.policy import MlpPolicy, RunningMeanStd
    dev = "cuda:0"
    pi = MlpPolicy(device=dev, seed=0)
    L = TrpoLearner(pi)
    ref = RunningMeanStd((56,), device=dev)
    g = torch.Generator(device=dev).manual_seed(n)
    for k in range(2):
        ob = (torch.randn((n, 56), device=dev, generator=g) * torch.linspace(0.3, 4.0, 56, device=dev) + 1.5 * k).contiguous()
        L._rms_update(ob)
        ref.update(ob)
        assert L._rms_scratch is not None                                     # the kernel ran, not the tensor ops
        assert float(pi.ob_rms.count) == float(ref.count)
        assert torch.allclose(pi.ob_rms.sum, ref.sum, rtol=1e-12, atol=1e-9) and torch.allclose(pi.ob_rms.sumsq, ref.sumsq, rtol=1e-12, atol=1e-9)
        assert torch.allclose(pi.ob_rms.mean, ref.mean, rtol=1e-6, atol=1e-7)
        # var = float32(sumsq / count) - mean^2 cancels (the reference's formula): an ulp of mean^2 is what a last-bit difference of the sums can move it by
        ulp = 4e-7 * (ref.mean ** 2 + (ref.sumsq / ref.count).float())
        assert bool(((pi.ob_rms.std ** 2 - ref.std ** 2).abs() <= ulp + 1e-9).all())


@pytest.mark.gpu
def test_learner_update_through_torch_autograd_matches_the_float64_restatement_on_gpu():
    """The path the kernels replace stays covered on the device (multi-backend fallback of the learner: pg_native=False)."""
    _check_against_golden(*_golden_update("cuda:0", pg_native=False))


@pytest.mark.gpu
@pytest.mark.parametrize("bs,nb", [(4096, 6), (37, 5), (9000, 2)])
def test_value_fit_with_the_epochs_filter_sums_up_front_equals_three_launches_per_minibatch(bs, nb):
    """dm_vf_fit_epoch(epoch_filter=1) — every minibatch's filter sums at once and their scan, then gradient + Adam per minibatch — against
    the three-launches-per-minibatch form: value parameters, Adam moments and the obs filter's state BIT for bit after two epochs
    (ragged minibatches; 9 000 samples: 282 blocks of partial gradients)."""
    from deepmimic_mujoco_amd.trpo import VF_KEYS, flat, TrpoLearner
    from deepmimic_mujoco_amd.policy import MlpPolicy
    torch.manual_seed(2)
    n = nb * bs + 3
    ob = torch.randn(n, 56, device="cuda:0") * 1.5 + 0.3; ret = torch.randn(n, device="cuda:0") * 2.0
    outs = []
    for one in (False, True):
        pi = MlpPolicy(device="cuda:0", seed=5)
        L = TrpoLearner(pi, vf_batch_size=bs, vf_iters=2, vf_graph=False, vf_native=True)
        L.vf_epoch_filter = one
        for k in (7, 8):
            inds = torch.randperm(n, generator=torch.Generator().manual_seed(k)).to("cuda:0")
            L._vf_native_epoch(ob, ret, inds, bs)
        torch.cuda.synchronize()
        r = pi.ob_rms
        outs.append((flat([pi.params[k].detach() for k in VF_KEYS]).clone(), L.vfadam.m.clone(), L.vfadam.v.clone(), r.sum.clone(), r.sumsq.clone(),
                     r.count.clone() if torch.is_tensor(r.count) else torch.tensor(float(r.count)), r.mean.clone(), r.std.clone()))
        assert L.vfadam.t == 2 * nb
    for x, y in zip(*outs):
        assert torch.equal(x, y)
    assert bool(torch.isfinite(outs[1][0]).all()) and float((outs[1][1]).abs().max()) > 0
