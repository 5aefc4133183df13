"""DM_OPT_STEP_QUEUE (include/dmenv.h): queued dm_batch_step calls run as one horizon launch — results, ordering rules and the
hand-over to / from every other entry point.  Reference semantics: T consecutive `DPEnv.step` calls (src/dp_env_v3.py:106-132)."""
import numpy as np
import pytest
import torch

from deepmimic_mujoco_amd import _abi as A
from deepmimic_mujoco_amd import DPVecEnv

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def _state(b):
    return (b.get(A.F_QPOS), b.get(A.F_QVEL), b.get(A.F_QACC_WARMSTART), b.get(A.F_FRAME_IDX), b.get(A.F_EPISODE), b.get(A.F_TIME), b.get(A.F_NEFC))


def _make(n, reward="imitation", queue=0, packed=True, pipeline=1, clip="walk"):
    env = DPVecEnv(n, motion=clip, device=0, reward=reward, autoreset="rsi", seed=3, packed=packed, frame_skip=1)
    env.batch.set_option(A.OPT_PIPELINE, pipeline)
    if queue:
        env.batch.set_option(A.OPT_STEP_QUEUE, queue)
    return env


@pytest.mark.parametrize("n,queue", [(642, 64), (4096, 20), (4096, 256), (37, 7)])
def test_queued_steps_equal_unqueued_packed_steps(n, queue):
    """T dm_batch_step calls with the queue on == the same calls with it off (four environments per wavefront both times), bit for bit:
    every step's observations, rewards and done flags, and the final state — including early terminations with RSI inside the queue and
    environments planted beyond the packed path's capacities (re-stepped inside their wave)."""
    from tests import helpers as H
    T = 45
    hi, hq, hv = H.many_row_states(40, 64, want=2)
    g = torch.Generator(device=DEV); g.manual_seed(11)
    ac = torch.randn((T, n, 28), generator=g, dtype=torch.float64, device=DEV) * 0.9
    outs = []
    for qd in (0, queue):
        env = _make(n, queue=qd)
        b = env.batch
        ob = torch.zeros((T, n, 56), dtype=torch.float64, device=DEV)
        rew = torch.zeros((T, n), dtype=torch.float64, device=DEV); dn = torch.zeros((T, n), dtype=torch.uint8, device=DEV)
        env.reset("rsi")
        q, v, f = b.get(A.F_QPOS), b.get(A.F_QVEL), b.get(A.F_FRAME_IDX)
        for e, k in ((1, 0), (n - 3, -1)):
            q[e], v[e], f[e] = hq[k], hv[k], hi[k]
        b.set_state(q, v, f)
        redo0 = b.redo_total()
        for t in range(T):
            b.step(ac[t], 1, (ob[t], rew[t], dn[t]))
        if qd:
            fl, st, pending = b.queue_stats()
            assert st + pending == T and 1 <= pending <= qd and st % qd == 0, (fl, st, pending)
        b.join(); b.sync()
        if qd:
            fl, st, pending = b.queue_stats()
            assert pending == 0 and st == T and fl == (T + qd - 1) // qd
        outs.append((ob.clone(), rew.clone(), dn.clone()) + _state(b) + (b.redo_total() - redo0,))
        env.close()
    x, y = outs
    assert int(y[2].sum()) > 0 and bool(torch.isfinite(y[0]).all())
    for i in range(3):
        assert torch.equal(x[i], y[i]), "row arrays differ (%d)" % i
    for i in range(3, 10):
        assert np.array_equal(x[i], y[i]), "final state differs (%d)" % i
    assert x[10] == y[10] > 0


def test_queue_against_the_oracle_and_other_entry_points_flush_it():
    """Queued steps against the CPU oracle (obs / reward 1e-9, done flags, cursors), with the entry points that must run the queue first
    in between: a field read, a host-pointer step, an option change.  The order of effects is the order of the calls."""
    from tests import helpers as H
    from oracle import oracle as O
    n, T = 64, 12
    env = _make(n, reward="v3-config", queue=8)
    b = env.batch
    om = H.oracle_model()
    mc = env.mocap
    idx = ((np.arange(n) * 5) % mc.data_config.shape[0]).astype(np.int32)
    q = mc.data_config[idx].copy(); v = mc.data_vel[idx].copy()
    b.set_option(A.OPT_AUTORESET, 0)
    b.set(A.F_QACC_WARMSTART, np.zeros((n, 34))); b.set(A.F_TIME, np.zeros(n))
    b.set_state(q, v, frame_idx=idx)
    ods = [O.Data(om) for _ in range(n)]
    for e in range(n):
        ods[e].reset(); ods[e].set_state(q[e], v[e])
    fidx = idx.astype(np.int64).copy()

    def oracle_step(a):
        o_all = np.zeros((n, 56)); r_all = np.zeros(n); d_all = np.zeros(n, dtype=np.uint8)
        for e in range(n):
            o, r, d, ic = ods[e].env_step(a[e], 1, 1, mc.data_config, int(fidx[e]), int(idx[e]))
            fidx[e] = ic
            o_all[e], r_all[e], d_all[e] = o, r, d
        return o_all, r_all, d_all

    rng = np.random.RandomState(5)
    acts = rng.randn(T + 3, n, 28) * 0.9
    ac = torch.as_tensor(acts, device=DEV)
    ob = torch.zeros((T, n, 56), dtype=torch.float64, device=DEV)
    rew = torch.zeros((T, n), dtype=torch.float64, device=DEV); dn = torch.zeros((T, n), dtype=torch.uint8, device=DEV)
    want_obs = np.zeros((T, n, 56)); want_rew = np.zeros((T, n)); want_done = np.zeros((T, n), dtype=np.uint8)
    for t in range(T):
        b.step(ac[t], 1, (ob[t], rew[t], dn[t]))
        want_obs[t], want_rew[t], want_done[t] = oracle_step(acts[t])
        if t == 4:                      # a field read in the middle of a queue: runs it, sees the state after step 4
            assert b.queue_stats()[2] == 5
            qp = b.get(A.F_QPOS)
            assert b.queue_stats()[2] == 0
            np.testing.assert_allclose(qp[:, 7:], want_obs[4][:, :28], rtol=0, atol=1e-9)
    assert b.queue_stats()[2] == (T - 5) % 8
    b.join(); b.sync()
    np.testing.assert_allclose(ob.cpu().numpy(), want_obs, rtol=0, atol=1e-9)
    np.testing.assert_allclose(rew.cpu().numpy(), want_rew, rtol=0, atol=1e-9)
    assert np.array_equal(dn.cpu().numpy(), want_done)
    assert np.array_equal(b.get(A.F_FRAME_IDX), fidx.astype(np.int32))
    # a host-pointer step after a queued one runs that one first
    b.step(ac[T], 1, (ob[0], rew[0], dn[0]))
    assert b.queue_stats()[2] == 1
    o_host, r_host, d_host = b.step(acts[T + 1])
    assert b.queue_stats()[2] == 0
    oracle_step(acts[T])
    o, r, d = oracle_step(acts[T + 1])
    np.testing.assert_allclose(o_host, o, rtol=0, atol=1e-9)
    np.testing.assert_allclose(r_host, r, rtol=0, atol=1e-9)
    # an option change applies AFTER what is queued: the queued step is still rewarded by the v3-config rule
    b.step(ac[T + 2], 1, (ob[2], rew[2], dn[2]))
    b.set_option(A.OPT_REWARD_MODE, 0)
    assert b.queue_stats()[2] == 0
    b.sync()
    o, r, d = oracle_step(acts[T + 2])
    np.testing.assert_allclose(rew[2].cpu().numpy(), r, rtol=0, atol=1e-9)
    env.close()


def test_buffer_reuse_degenerates_to_step_by_step_and_closed_loops_stay_correct():
    """A caller that hands the SAME tensors to consecutive calls (and, by the contract, joins before it reads them) gets plain step-by-step
    behaviour from a queued batch: each call first runs the one before."""
    n, T = 256, 10
    g = torch.Generator(device=DEV); g.manual_seed(2)
    acs = torch.randn((T, n, 28), generator=g, dtype=torch.float64, device=DEV) * 0.9
    res = []
    for qd in (0, 32):
        env = _make(n, reward="alive", queue=qd)
        b = env.batch
        env.reset("rsi")
        act = torch.zeros((n, 28), dtype=torch.float64, device=DEV)
        out = (torch.zeros((n, 56), dtype=torch.float64, device=DEV), torch.zeros(n, dtype=torch.float64, device=DEV), torch.zeros(n, dtype=torch.uint8, device=DEV))
        trace = []
        for t in range(T):
            act.copy_(acs[t])
            b.step(act, 1, out)
            b.join()
            trace.append(out[0].clone())
        b.sync()
        if qd:
            assert b.queue_stats()[:2] == (T, T)
        res.append(torch.stack(trace))
        env.close()
    assert torch.equal(res[0], res[1])


def test_vecenv_facade_is_unaffected_by_a_queue():
    """`DPVecEnv.step` hands out fresh tensors per call, so calls queue; reading the result without a join is the caller's bug under the
    pipelined contract — the facade's `step_wait` therefore joins when a queue is configured."""
    n = 128
    g = torch.Generator(device=DEV); g.manual_seed(3)
    acs = torch.randn((6, n, 28), generator=g, dtype=torch.float64, device=DEV) * 0.9
    outs = []
    for qd in (0, 16):
        env = _make(n, reward="alive", queue=qd)
        env.reset("rsi")
        tr = [env.step(acs[t])[0].clone() for t in range(6)]
        torch.cuda.synchronize()
        outs.append(torch.stack(tr))
        env.close()
    assert torch.equal(outs[0], outs[1])


def test_changing_the_pipeline_depth_between_packed_steps_keeps_the_redo_counters_clean():
    """ADVICE (round 3): packed steps at pipeline depth 4, then an odd number at depth 2, then 4 again used to leave a stale redo counter
    for the sub-batches beyond the smaller depth — some environments were then stepped twice in one call.  Against one launch per step."""
    from tests import helpers as H
    n, T = 520, 14
    hi, hq, hv = H.many_row_states(40, 64, want=4)
    g = torch.Generator(device=DEV); g.manual_seed(9)
    ac = torch.randn((T, n, 28), generator=g, dtype=torch.float64, device=DEV) * 0.9
    depths = [4, 4, 4, 2, 2, 2, 4, 4, 4, 2, 4, 2, 4, 4]
    outs = []
    for vary in (False, True):
        env = _make(n, reward="alive", packed=True)
        b = env.batch
        env.reset("rsi")
        ob = torch.zeros((T, n, 56), dtype=torch.float64, device=DEV)
        rew = torch.zeros((T, n), dtype=torch.float64, device=DEV); dn = torch.zeros((T, n), dtype=torch.uint8, device=DEV)
        for t in range(T):
            # environments beyond the packed path's capacities in every sub-batch, every few steps: the redo lists are in use throughout
            if t % 3 == 0:
                q, v, f = b.get(A.F_QPOS), b.get(A.F_QVEL), b.get(A.F_FRAME_IDX)
                for k, e in enumerate((3, n // 4 + 5, n // 2 + 7, 3 * n // 4 + 9)):
                    q[e], v[e], f[e] = hq[k % len(hq)], hv[k % len(hv)], hi[k % len(hi)]
                b.set_state(q, v, f)
            b.set_option(A.OPT_PIPELINE, depths[t] if vary else 1)
            b.step(ac[t], 1, (ob[t], rew[t], dn[t]))
        b.join(); b.sync()
        outs.append((ob.clone(), rew.clone(), dn.clone()) + _state(b) + (b.redo_total(),))
        env.close()
    x, y = outs
    assert x[10] > 0
    for i in range(3):
        assert torch.equal(x[i], y[i]), "row arrays differ (%d)" % i
    for i in range(3, 10):
        assert np.array_equal(x[i], y[i]), "final state differs (%d)" % i
    assert x[10] == y[10]


def test_dpvecenv_step_queue_argument():
    """`DPVecEnv(step_queue=Q)` = DM_OPT_PACKED + DM_OPT_STEP_QUEUE on its batch: raw `batch.step` calls queue, the facade's `step` still returns finished results."""
    n = 96
    env = DPVecEnv(n, motion="walk", device=0, reward="alive", autoreset="rsi", seed=1, step_queue=16)
    assert env.packed and env.batch.options[A.OPT_STEP_QUEUE] == 16
    env.reset("rsi")
    g = torch.Generator(device=DEV); g.manual_seed(4)
    acs = torch.randn((5, n, 28), generator=g, dtype=torch.float64, device=DEV) * 0.5
    outs = [(torch.zeros((n, 56), dtype=torch.float64, device=DEV), torch.zeros(n, dtype=torch.float64, device=DEV), torch.zeros(n, dtype=torch.uint8, device=DEV)) for _ in range(4)]
    for t in range(4):
        env.batch.step(acs[t], 1, outs[t])
    assert env.batch.queue_stats() == (0, 0, 4)
    ob, rew, done, infos = env.step(acs[4])                 # joins: the four queued steps ran first, then this one
    assert env.batch.queue_stats()[2] == 0 and env.batch.queue_stats()[1] == 5
    torch.cuda.synchronize()
    assert bool(torch.isfinite(ob).all()) and float(rew.min()) == 1.0 and len(infos) == n
    assert float(outs[3][0].abs().sum()) > 0
    env.close()


def test_a_call_that_overlaps_a_queued_buffer_in_any_role_runs_the_queue_first():
    """dmenv.h DM_OPT_STEP_QUEUE: "overlaps" is a byte-range test over all four buffers of every queued call (round 4 compared base pointers per role).
    Views that start elsewhere in a queued call's tensors and a buffer that changes its role are handed in; each such call must flush (queue_stats),
    and the results equal the same calls with the queue off."""
    n = 512
    g = torch.Generator(device=DEV); g.manual_seed(2)
    ac = torch.randn((6, n, 28), generator=g, dtype=torch.float64, device=DEV) * 0.9
    res = []
    for qd in (0, 16):
        env = _make(n, reward="alive", queue=qd)
        b = env.batch
        env.reset("rsi")
        big_o = torch.zeros((2 * n + 8, 56), dtype=torch.float64, device=DEV)     # overlapping [n, 56] windows live in here
        big_r = torch.zeros(2 * n, dtype=torch.float64, device=DEV); big_d = torch.zeros(2 * n, dtype=torch.uint8, device=DEV)
        o1 = torch.zeros((n, 56), dtype=torch.float64, device=DEV); r1 = torch.zeros(n, dtype=torch.float64, device=DEV); d1 = torch.zeros(2 * n, dtype=torch.uint8, device=DEV)
        fl = []
        flushes = lambda: b.queue_stats()[0] if qd else 0
        b.step(ac[0], 1, (big_o[:n], big_r[:n], big_d[:n])); fl.append(flushes())
        b.step(ac[1], 1, (big_o[8:n + 8], big_r[n:], big_d[n:])); fl.append(flushes())                        # obs window overlaps step 0's at another base address
        b.step(ac[2], 1, (o1, big_r[n // 2:n // 2 + n], d1[:n])); fl.append(flushes())                        # reward window straddles step 1's reward buffer
        obs3 = big_o[n + 8:2 * n + 8]
        b.step(ac[3], 1, (obs3, r1, d1[n // 2:n // 2 + n])); fl.append(flushes())                             # done window overlaps step 2's
        b.step(obs3.view(-1)[:n * 28].view(n, 28), 1, (big_o[:n], big_r[:n], big_d[:n])); fl.append(flushes())  # a queued call's OUTPUT bytes as this call's action
        b.join(); b.sync()
        res.append((fl, big_o.clone(), big_r.clone(), big_d.clone(), o1.clone(), r1.clone(), d1.clone()) + _state(b))
        env.close()
    x, y = res
    assert y[0] == [0, 1, 2, 3, 4], y[0]            # every overlapping call ran what was queued before it was queued itself
    for i in range(1, 7):
        assert torch.equal(x[i], y[i]), i
    for i in range(7, len(x)):
        assert np.array_equal(x[i], y[i]), i
