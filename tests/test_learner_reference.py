"""The learner's helpers against outputs of the REFERENCE's own code (tests/golden/learner_ref_golden.npz, written by
tests/golden/gen/make_learner_fixture.py: src/cg.py, src/dataset.py and src/utils/math_util.py imported as they are; add_vtarg_and_adv of
src/trpo.py:83-94, MpiAdam of src/mpi_adam.py:6-35 and RunningMeanStd of src/utils/misc_util.py:32-70 cut out of their modules' syntax trees and executed
against one-rank / numpy stand-ins).  Rounds 3-4 compared these helpers with restatements typed into the tests; the restatements stay where they check
more than the fixture holds, the fixture is what pins them."""
import os

import numpy as np
import pytest
import torch

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "learner_ref_golden.npz"))


def test_cg_equals_the_references_cg():
    from deepmimic_mujoco_amd.trpo import cg
    A = torch.from_numpy(G["cg_A"]); b = torch.from_numpy(G["cg_b"])
    for iters in (1, 3, 10, 60):
        for sync_free in (False, True):
            x = cg(lambda p: A @ p, b.clone(), cg_iters=iters, sync_free=sync_free)
            if iters <= 10:
                assert np.allclose(x.numpy(), G["cg_x_%d" % iters], rtol=1e-9, atol=1e-11), iters
            else:       # run to the residual test (r.r < 1e-10): both have solved the system; which iteration trips the test depends on the last bits
                sol = np.linalg.solve(G["cg_A"], G["cg_b"])
                assert np.abs(x.numpy() - sol).max() < 1e-4 and np.abs(G["cg_x_%d" % iters] - sol).max() < 1e-4
    for sync_free in (False, True):                                           # the early break (residual_tol) lands on the same iterate
        x = cg(lambda p: A @ p, b.clone(), cg_iters=60, residual_tol=1e-3, sync_free=sync_free)
        assert np.allclose(x.numpy(), G["cg_x_tol"], rtol=1e-9, atol=1e-11)
    assert not np.allclose(G["cg_x_tol"], G["cg_x_60"], rtol=1e-6)              # (the break is in the fixture)
    A32 = A.to(torch.float32); b32 = b.to(torch.float32)                        # float32, as the trainer runs it
    x = cg(lambda p: A32 @ p, b32.clone(), cg_iters=10)
    assert x.dtype == torch.float32 and np.allclose(x.numpy(), G["cg_x32_10"], rtol=2e-4, atol=2e-5)


def test_explained_variance_equals_the_references():
    from deepmimic_mujoco_amd.trpo import explained_variance
    y = torch.from_numpy(G["ev_y"]); yp = torch.from_numpy(G["ev_ypred"])
    got = [float(explained_variance(yp, y)), float(explained_variance(torch.zeros_like(y), y))]
    assert np.allclose(got, G["ev"][:2], rtol=1e-5)
    assert np.isnan(G["ev"][2]) and np.isnan(float(explained_variance(yp, torch.ones_like(y))))     # Var[y] = 0 -> nan, as the reference


def test_gae_equals_the_references_add_vtarg_and_adv():
    from deepmimic_mujoco_amd.rollout import add_vtarg_and_adv
    seg = {"rew": torch.from_numpy(G["gae_rew"]), "vpred": torch.from_numpy(G["gae_vpred"]), "new": torch.from_numpy(G["gae_new"]),
           "nextvpred": torch.from_numpy(G["gae_nextvpred"])}
    gamma, lam = (float(v) for v in G["gae_gamma_lam"])
    add_vtarg_and_adv(seg, gamma, lam)
    # (the reference accumulates in Python floats — float64 — and stores float32; the batched form runs in float32: agreement to float32 rounding of a
    #  256-step recursion whose terms reach |adv| ~ 50)
    assert np.allclose(seg["adv"].numpy(), G["gae_adv"], rtol=2e-5, atol=2e-4)
    assert np.allclose(seg["tdlamret"].numpy(), G["gae_tdlamret"], rtol=2e-5, atol=2e-4)
    assert float(np.abs(G["gae_adv"]).max()) > 10


def test_mpi_adam_equals_the_references_update_rule():
    from deepmimic_mujoco_amd.trpo import MpiAdam
    th0 = G["adam_theta0"]
    w = torch.from_numpy(th0[:35].reshape(7, 5).copy()); b = torch.from_numpy(th0[35:].copy())
    opt = MpiAdam([w, b], epsilon=1e-8)
    for k, g in enumerate(G["adam_grads"]):
        opt.update(torch.from_numpy(g), 1e-3)
        th = np.concatenate([w.numpy().ravel(), b.numpy().ravel()])
        assert np.allclose(th, G["adam_theta"][k], rtol=1e-6, atol=1e-7), k
    assert np.allclose(opt.m.numpy(), G["adam_m"], rtol=1e-5, atol=1e-9) and np.allclose(opt.v.numpy(), G["adam_v"], rtol=1e-5, atol=1e-12)
    assert float(np.abs(G["adam_theta"][-1] - th0).max()) > 5e-3                # twelve steps of 1e-3 each moved the parameters


def test_running_mean_std_equals_the_references():
    from deepmimic_mujoco_amd.policy import RunningMeanStd
    rms = RunningMeanStd((56,))
    assert np.allclose(rms.mean.numpy(), G["rms_mean"][0]) and np.allclose(rms.std.numpy(), G["rms_std"][0])
    for k in range(3):
        rms.update(torch.from_numpy(G["rms_x%d" % k]))
        assert np.allclose(rms.mean.numpy(), G["rms_mean"][k + 1], rtol=1e-5, atol=1e-6), k
        assert np.allclose(rms.std.numpy(), G["rms_std"][k + 1], rtol=1e-5, atol=1e-6), k
    r2 = RunningMeanStd((4,))                                                   # nearly constant data: the variance floor, std = sqrt(1e-2)
    r2.update(torch.from_numpy(G["rms_floor_x"]))
    assert np.allclose(r2.mean.numpy(), G["rms_floor_mean"], rtol=1e-5, atol=1e-6) and np.allclose(r2.std.numpy(), G["rms_floor_std"], rtol=1e-5)
    assert float(G["rms_floor_std"].max()) == pytest.approx(0.1, rel=1e-6) and float(G["rms_std"][0][0]) == pytest.approx(1.0)


def test_value_fit_minibatches_have_the_references_structure():
    """src/trpo.py:288-296 + src/dataset.py:50-60: vf_iters epochs, each one fresh permutation of all samples cut into batches of 128 with the partial
    tail dropped.  The fixture holds the reference's batches for a seeded numpy stream (the order itself is numpy's global Mersenne twister: not
    reproduced on the device); the learner's plan has the same structure."""
    seed, N, bs = (int(v) for v in G["iterbatches_seed"])
    order = G["iterbatches_order"]
    nb = N // bs
    assert order.shape == (3 * nb, bs)
    for e in range(3):
        ep = order[e * nb:(e + 1) * nb].ravel()
        assert len(set(ep.tolist())) == nb * bs and ep.min() >= 0 and ep.max() < N          # distinct samples: a permutation's head
    assert not np.array_equal(order[:nb], order[nb:2 * nb])
    from deepmimic_mujoco_amd.trpo import TrpoLearner
    from deepmimic_mujoco_amd.policy import MlpPolicy
    L = TrpoLearner(MlpPolicy(seed=0), vf_batch_size=bs, vf_iters=3, seed=3)
    perms = [L._next_perm(N, torch.device("cpu")) for _ in range(3)]
    for p in perms:
        assert sorted(p.tolist()) == list(range(N))
    assert not torch.equal(perms[0], perms[1])


@pytest.mark.gpu
def test_device_kernels_equal_the_references_outputs():
    """The same fixtures on the MI355X: `dm_gae` (k_gae: one launch for the segment) against add_vtarg_and_adv, `dm_rms_update` (one launch: sums, counts,
    mean, std) against RunningMeanStd.update + its mean / std expressions, the conjugate-gradient loop on device tensors (both forms of the residual
    test) against src/cg.py, and MpiAdam on device tensors against src/mpi_adam.py's rule."""
    import ctypes as C
    from deepmimic_mujoco_amd import _abi as A
    from deepmimic_mujoco_amd.rollout import add_vtarg_and_adv
    from deepmimic_mujoco_amd.policy import RunningMeanStd
    from deepmimic_mujoco_amd.trpo import cg, MpiAdam
    dev = torch.device("cuda", 0)
    # GAE
    seg = {"rew": torch.from_numpy(G["gae_rew"]).to(dev), "vpred": torch.from_numpy(G["gae_vpred"]).to(dev), "new": torch.from_numpy(G["gae_new"]).to(dev),
           "nextvpred": torch.from_numpy(G["gae_nextvpred"]).to(dev)}
    add_vtarg_and_adv(seg, *(float(v) for v in G["gae_gamma_lam"]))
    assert seg["adv"].is_cuda
    assert np.allclose(seg["adv"].cpu().numpy(), G["gae_adv"], rtol=2e-5, atol=2e-4) and np.allclose(seg["tdlamret"].cpu().numpy(), G["gae_tdlamret"], rtol=2e-5, atol=2e-4)
    # obs filter: dm_rms_update on the device-resident float64 sums
    L = A.load()
    rms = RunningMeanStd((56,), device=dev)
    scratch = torch.empty(int(L.dm_rms_scratch_bytes()), dtype=torch.uint8, device=dev)
    p = lambda x: C.c_void_p(x.data_ptr())
    for k in range(3):
        ob = torch.from_numpy(G["rms_x%d" % k]).to(dev).contiguous()
        cnt = rms.count if torch.is_tensor(rms.count) else None
        if cnt is None:
            pytest.skip("RunningMeanStd keeps its count on the host in this configuration")
        A.check(L.dm_rms_update(p(ob), int(ob.shape[0]), p(rms.sum), p(rms.sumsq), p(rms.count), p(rms.mean), p(rms.std), p(scratch),
                                C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), L)
        torch.cuda.synchronize()
        assert np.allclose(rms.mean.cpu().numpy(), G["rms_mean"][k + 1], rtol=1e-5, atol=1e-6), k
        assert np.allclose(rms.std.cpu().numpy(), G["rms_std"][k + 1], rtol=1e-5, atol=1e-6), k
    assert np.allclose(rms.sum.cpu().numpy(), G["rms_sum"], rtol=1e-12) and np.allclose(rms.sumsq.cpu().numpy(), G["rms_sumsq"], rtol=1e-12)
    assert float(rms.count) == pytest.approx(float(G["rms_count"]), rel=1e-15)
    # CG on device tensors
    Ad = torch.from_numpy(G["cg_A"]).to(dev); bd = torch.from_numpy(G["cg_b"]).to(dev)
    for sync_free in (False, True):
        x = cg(lambda q: Ad @ q, bd.clone(), cg_iters=10, sync_free=sync_free)
        assert np.allclose(x.cpu().numpy(), G["cg_x_10"], rtol=1e-8, atol=1e-10)
        x = cg(lambda q: Ad @ q, bd.clone(), cg_iters=60, residual_tol=1e-3, sync_free=sync_free)
        assert np.allclose(x.cpu().numpy(), G["cg_x_tol"], rtol=1e-8, atol=1e-10)
    # Adam on device tensors
    th0 = G["adam_theta0"]
    w = torch.from_numpy(th0[:35].reshape(7, 5).copy()).to(dev); b = torch.from_numpy(th0[35:].copy()).to(dev)
    opt = MpiAdam([w, b], epsilon=1e-8)
    for k, g in enumerate(G["adam_grads"]):
        opt.update(torch.from_numpy(g).to(dev), 1e-3)
    th = np.concatenate([w.cpu().numpy().ravel(), b.cpu().numpy().ravel()])
    assert np.allclose(th, G["adam_theta"][-1], rtol=1e-6, atol=1e-7)
