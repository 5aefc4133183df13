"""The HIP kernel source (deepmimic_mujoco_amd/csrc/env_*.h), executed lane-for-lane on the fibre wave testbench
(tests/emu), against the CPU oracle.  Runs in the GPU-less container; the same comparisons run on the real device
in tests/test_gpu_parity.py."""
import numpy as np
import pytest

from deepmimic_mujoco_amd import _abi as A
from tests import helpers as H
from tests.emu.emu import EmuBatch


def make(n, flags=0):
    mc = H.mocap()
    return EmuBatch(H.compiled_model(), mc.data_config, mc.data_vel, n, flags)


def test_forward_stages_match_oracle_on_testbench():
    n = 12
    worst = H.compare_forward(make(n), H.oracle_model(), *H.varied_states(n, seed=3))
    assert max(worst.values()) < 1e-11


def test_rollout_matches_oracle_on_testbench():
    n = 6
    idx, q, v, _w, _c = H.varied_states(n, seed=5)
    worst, ndone = H.compare_rollout(make(n), H.oracle_model(), idx, q, v, steps=15, seed=1)
    assert worst < 1e-10


def test_no_contact_config_on_testbench():
    n = 3
    idx, q, v, _w, _c = H.varied_states(n, seed=7)
    b = make(n, A.FLAG_NO_CONTACT | A.FLAG_NO_LIMIT)
    H.compare_rollout(b, H.oracle_model(enable_contact=0, enable_limit=0), idx, q, v, steps=8, seed=2)
    assert np.all(b.get(A.F_NEFC) == 0)


@pytest.mark.parametrize("mode", [1, 2])
def test_reward_modes_on_testbench(mode):
    n = 2
    b = make(n)
    b.set_option(A.OPT_REWARD_MODE, mode)
    idx, q, v, _w, _c = H.varied_states(n, seed=11)
    H.compare_rollout(b, H.oracle_model(), idx, q, v, steps=5, seed=3, reward_mode=mode)


def test_row_capacity_overflow_is_reported_and_mirrors_oracle():
    """A humanoid pressed flat into the floor produces more than 63 constraint rows: contacts past the on-chip capacity
    are dropped in list order, status bit 0 is set, and the kept rows still match the oracle with the same cap."""
    mc = H.mocap()
    q = mc.data_config[0:1].copy(); v = np.zeros((1, 34))
    q[0, 2] = 0.02; q[0, 3:7] = [np.sqrt(0.5), 0, np.sqrt(0.5), 0]; q[0, 7:] = 0.0
    b = make(1)
    ws = np.zeros((1, 34)); ctrl = np.zeros((1, 28))
    H.compare_forward(b, H.oracle_model(), np.zeros(1, dtype=np.int32), q, v, ws, ctrl)
    assert b.get(A.F_NEFC)[0] <= 64
    if b.get(A.F_NCON)[0] * 4 > 64:
        assert b.get(A.F_STATUS)[0] & 1


def test_reset_and_autoreset_on_testbench():
    n = 4
    mc = H.mocap()
    b = make(n)
    b.set_option(A.OPT_SEED, 9)
    b.reset(0, 1)
    fi = b.get(A.F_FRAME_IDX)
    assert np.array_equal(b.get(A.F_QPOS), mc.data_config[fi]) and np.array_equal(b.get(A.F_QVEL), mc.data_vel[fi])
    assert np.all(b.get(A.F_EPISODE) == 1)
    b2 = make(n); b2.set_option(A.OPT_SEED, 9); b2.set_option(A.OPT_ENV_OFFSET, 0); b2.reset(0, 1)
    assert np.array_equal(b2.get(A.F_FRAME_IDX), fi), "reset RNG must be a pure function of (seed, env, episode)"
    b3 = make(2); b3.set_option(A.OPT_SEED, 9); b3.set_option(A.OPT_ENV_OFFSET, 2); b3.reset(0, 1)
    assert np.array_equal(b3.get(A.F_FRAME_IDX), fi[2:]), "sharding must not change per-env RNG streams"
    b.reset(1, 1)
    q = b.get(A.F_QPOS)
    assert np.all(np.abs(q - H.compiled_model().qpos0) <= 0.01) and np.all(b.get(A.F_TIME) == 0)
    # auto-reset: start below the termination height so the first step reports done and re-initialises
    b.set_option(A.OPT_AUTORESET, 1)
    qq = mc.data_config[[0, 1, 2, 3]].copy(); qq[:2, 2] = 0.3
    b.set_state(qq, mc.data_vel[[0, 1, 2, 3]].copy())
    obs, rew, done = b.step(np.zeros((n, 28)))
    assert list(done[:2]) == [1, 1]
    fi = b.get(A.F_FRAME_IDX)
    for e in np.nonzero(done)[0]:
        assert np.array_equal(b.get(A.F_QPOS)[e], mc.data_config[fi[e]]) and b.get(A.F_TIME)[e] == 0
        assert np.array_equal(obs[e][:28], mc.data_config[fi[e]][7:])


def test_action_front_ends_on_testbench():
    """P-control (src/env_torque_test.py:20) and PD (src/mujoco/setting_states.py:207-226) action front-ends: the ctrl the
    kernel stores must equal the host formula, and the step must equal an oracle step driven with that ctrl."""
    from oracle import oracle as O
    mc = H.mocap()
    n = 2
    idx, q, v, _w, _c = H.varied_states(n, seed=21)
    kp = np.load(H.GOLDEN + "/env_logic_golden.npz")["kp"]; kd = np.load(H.GOLDEN + "/env_logic_golden.npz")["kd"]
    rng = np.random.RandomState(3)
    for mode in (1, 2):
        b = make(n)
        b.set_option(A.OPT_ACTION_MODE, mode)
        b.set(A.F_QACC_WARMSTART, np.zeros((n, 34))); b.set(A.F_TIME, np.zeros(n))
        b.set_state(q, v, frame_idx=idx)
        a = rng.randn(n, 28) * 0.1
        obs, rew, done = b.step(a)
        if mode == 1:
            expect = a + 0.8 * (mc.data_config[idx][:, 7:] - q[:, 7:])
        else:
            expect = a + kp * (mc.data_config[idx][:, 7:] - q[:, 7:]) + kd * (mc.data_vel[idx][:, 6:] - v[:, 6:])
        assert np.abs(b.get(A.F_CTRL) - expect).max() < 1e-12
        om = H.oracle_model()
        for e in range(n):
            od = O.Data(om); od.reset(); od.set_state(q[e], v[e])
            o, r, d, _ = od.env_step(expect[e])
            assert H.rel_err(obs[e], o) < 1e-10


def test_capsule_box_contacts_on_testbench():
    """Shin/thigh capsule against the opposite foot box (seeded poses found offline, tests/golden/capsule_box_poses.npy)."""
    qs = np.load(H.GOLDEN + "/capsule_box_poses.npy")[:3]
    n = len(qs)
    b = make(n)
    H.compare_forward(b, H.oracle_model(), np.zeros(n, dtype=np.int32), qs, np.zeros((n, 34)), np.zeros((n, 34)), np.zeros((n, 28)))
    cg = b.get(A.F_CONTACT_GEOMS)
    assert all(any((c[0] in (10, 11, 13, 14)) and (c[1] in (12, 15)) for c in cg[e] if c[0] >= 0) for e in range(n))


def test_capsule_through_the_box_interior_is_deterministic_on_testbench():
    """A shin capsule whose axis runs THROUGH the inside of the other foot's box (deep penetration; found by tools/fuzz_parity.py):
    the segment-to-box distance is zero on a whole interval there and the derivative's computed value at the interval's ends is +-1 ulp
    with a rounding-dependent sign.  The routine takes the plateau's point nearest the capsule's centre, so kernel and oracle (different
    instruction sequences, FMA contraction on one side only) still agree on the contact — before, they picked opposite ends (J off by
    0.78); the plateau's MIDDLE would tie the closest-face choice when the axis enters and leaves through opposite faces (pose 1)."""
    qs = np.load(H.GOLDEN + "/capsule_box_deep_poses.npy")
    n = len(qs)
    b = make(n)
    H.compare_forward(b, H.oracle_model(), np.zeros(n, dtype=np.int32), qs, np.zeros((n, 34)), np.zeros((n, 34)), np.zeros((n, 28)))
    cg = b.get(A.F_CONTACT_GEOMS)
    assert any(tuple(c) == (11, 15) for c in cg[0])                     # right shin capsule vs left foot box


def test_box_box_contacts_on_testbench():
    """Foot box against foot box (face contacts with 2-4 clipped vertices and edge-edge contacts; seeded poses)."""
    qs = np.load(H.GOLDEN + "/box_box_poses.npy")[[0, 1, 2]]
    n = len(qs)
    b = make(n)
    H.compare_forward(b, H.oracle_model(), np.zeros(n, dtype=np.int32), qs, np.zeros((n, 34)), np.zeros((n, 34)), np.zeros((n, 28)))
    cg = b.get(A.F_CONTACT_GEOMS)
    assert all(any(tuple(c) == (12, 15) for c in cg[e]) for e in range(n))


def test_overflow_strip_rows_match_oracle_and_the_all_register_tier():
    """Evaluations with more rows than the register tier holds (32 on the testbench) keep the remaining columns of A in the
    per-env memory strip: results must match the oracle and be bit-identical to the all-register (64-column) tier."""
    idx, q, v = H.many_row_states(32, 64, want=4)
    n = len(q)
    b = make(n)
    worst, _ = H.compare_rollout(b, H.oracle_model(), idx, q, v, steps=4, seed=4, action_scale=0.3)
    assert worst < 1e-10
    outs = []
    for tier in (1, 0):
        bb = make(n)
        bb.set_option(102, tier)
        bb.set(A.F_QACC_WARMSTART, np.zeros((n, 34))); bb.set_state(q, v, frame_idx=idx)
        rng = np.random.RandomState(0)
        o = [bb.step(rng.randn(n, 28) * 0.3)[0].copy() for _ in range(3)]
        assert bb.get(A.F_NEFC).max() > 32
        outs.append(np.stack(o))
    assert np.array_equal(outs[0], outs[1])


def test_pgs_guarded_replay_path_gives_identical_results():
    """Option 103 sends every PGS sweep through the guarded replay (columns of A parked in the memory strip, cost-change
    test in place): results must be bit-identical to the speculative sweep, which only replays when a row trips the test."""
    idx, q, v, _w, _c = H.varied_states(6, seed=17)
    i2, q2, v2 = H.many_row_states(32, 64, want=2)
    idx = np.concatenate([idx, i2]); q = np.concatenate([q, q2]); v = np.concatenate([v, v2])
    n = len(q)
    outs = []
    for force in (0, 1):
        b = make(n)
        b.set_option(103, force)
        b.set(A.F_QACC_WARMSTART, np.zeros((n, 34))); b.set_state(q, v, frame_idx=idx)
        rng = np.random.RandomState(1)
        o = [b.step(rng.randn(n, 28) * 0.5)[0].copy() for _ in range(3)]
        outs.append((np.stack(o), b.get(A.F_SOLVER_ITER).copy(), b.get(A.F_QACC_WARMSTART).copy()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][2], outs[1][2])
    assert outs[0][1].max() > 1


def test_packed_slots_match_oracle_on_the_rowless_model():
    """slot_kernel.h / slot_step.h (four environments per wavefront, one 16-lane row each) executed on the fibre testbench: rollouts of
    the contact-free, limit-free model (BASELINE.json configs[1]) against the oracle in the three reward modes the packed epilogue
    covers; 6 envs = one full wave + one wave with two spare slots."""
    from tests.emu.emu import EmuBatch
    mc = H.mocap()
    n = 6
    flags = A.FLAG_NO_CONTACT | A.FLAG_NO_LIMIT
    idx, q, v, _ws, _c = H.varied_states(n, seed=7)
    om = H.oracle_model(enable_contact=0, enable_limit=0)
    for mode in (0, 1, 2):
        b = EmuBatch(H.compiled_model(), mc.data_config, mc.data_vel, n, flags)
        b.set_option(A.OPT_PACKED, 1); b.set_option(A.OPT_REWARD_MODE, mode)
        worst, _nd = H.compare_rollout(b, om, idx, q, v, steps=5, seed=2, reward_mode=mode, n_substeps=2 if mode == 1 else 1)
        assert worst < 1e-12


def test_packed_slots_match_oracle_with_contacts_and_limits():
    """The full model on the packed path (collision compaction, per-contact staging, DPP-row constraint solve with one and two row
    sets, frozen sweeps of converged environments) against the oracle on the fibre testbench; the varied states include poses
    that exceed the packed path's capacities — those environments come back through the one-env kernel (redo list)."""
    from tests.emu.emu import EmuBatch
    mc = H.mocap()
    n = 8
    idx, q, v, _ws, _c = H.varied_states(n, seed=5)
    b = EmuBatch(H.compiled_model(), mc.data_config, mc.data_vel, n, 0)
    b.set_option(A.OPT_PACKED, 1)
    worst, _nd = H.compare_rollout(b, H.oracle_model(), idx, q, v, steps=4, seed=1)
    assert worst < 1e-11
    assert 0 < b.redo_total() < 4 * n and b.get(A.F_NEFC).max() > 16        # both row-set instantiations and the redo path ran


@pytest.mark.parametrize("mode", [1, 3])
def test_packed_horizon_in_one_wave_equals_step_by_step(mode):
    """(mode 3, the 5-term imitation reward: a step inside a horizon skips its first evaluation's position stage where the slots still hold the
    kinematics pass the previous step's reward ended with — `kin_carry` — and must not where a reset or an in-wave re-step intervened.)
    slot_step.h slot_rollout (k_rollout_packed: T steps of a wave's four environments without leaving the wave, an environment past
    the packed path's capacities re-stepped in place by the one-env code whose LDS ALIASES the slots') against the per-step routing
    (k_step_packed + k_step_redo) on the fibre testbench: every row of obs / reward / done and the final state bit for bit, with
    auto-reset on and redo events inside the horizon."""
    from tests.emu.emu import EmuBatch
    mc = H.mocap()
    n, T = 10, 4 if mode == 1 else 7              # two full waves + one wave with two spare slots
    imit = None
    if mode == 3:
        from deepmimic_mujoco_amd.imitation import ImitationSpec
        imit = ImitationSpec(H.compiled_model()).table_for(mc)
    idx, q, v, _ws, _c = H.varied_states(n, seed=5)
    hi, hq, hv = H.many_row_states(40, 64, want=2)          # more rows than a slot holds: these are re-stepped by the one-env code
    idx[1], q[1], v[1] = hi[0], hq[0], hv[0]
    idx[9], q[9], v[9] = hi[-1], hq[-1], hv[-1]
    acts = np.random.RandomState(3).randn(T, n, 28) * 0.9
    outs = []
    for horizon in (False, True):
        b = EmuBatch(H.compiled_model(), mc.data_config, mc.data_vel, n, 0, imitation=imit)
        b.set_option(A.OPT_PACKED, 1); b.set_option(A.OPT_REWARD_MODE, mode); b.set_option(A.OPT_AUTORESET, 1); b.set_option(A.OPT_SEED, 9)
        b.set(A.F_QACC_WARMSTART, np.zeros((n, 34))); b.set(A.F_TIME, np.zeros(n))
        b.set_state(q, v, frame_idx=idx)
        if horizon:
            obs, rew, done = b.rollout(acts)
        else:
            obs = np.zeros((T, n, 56)); rew = np.zeros((T, n)); done = np.zeros((T, n), dtype=np.uint8)
            for t in range(T):
                b.step(acts[t], 1, out=(obs[t], rew[t], done[t]))
        outs.append((obs, rew, done, b.get(A.F_QPOS), b.get(A.F_QVEL), b.get(A.F_QACC_WARMSTART), b.get(A.F_FRAME_IDX), b.get(A.F_EPISODE), b.get(A.F_TIME), b.redo_total()))
    for x, y in zip(outs[0][:-1], outs[1][:-1]):
        assert np.array_equal(x, y)
    assert outs[0][-1] == outs[1][-1] > 0, "the horizon must contain capacity overflows (%d / %d)" % (outs[0][-1], outs[1][-1])
    if mode == 3:
        assert int(outs[1][2].sum()) > 0, "the imitation horizon must contain auto-resets (kin_carry invalidated)"


def test_self_ordering_launches_dispatch_a_permutation_longest_first():
    """env_step.h order_ticket / dispatch_env (per-step launches order themselves; no k_order between two steps): whatever the arrival order of the
    previous launch's envs, the next launch's waves find every env of the part exactly once, cost keys descending — one env per wave and four per
    wave — and the launch's first workgroup clears the counters of the launch after it.  Without tickets: the stored order, else the identity."""
    from tests.emu import emu
    rng = np.random.RandomState(11)
    n = 300
    for first, count in ((0, 300), (64, 131), (297, 3), (10, 1)):
        nefc = rng.randint(0, 70, n); it = rng.randint(0, 51, n)
        arrival = first + rng.permutation(count)
        o1, o4 = emu.dispatch(n, first, count, nefc, it, arrival)
        key = np.clip(nefc + (it >> 2), 0, 63)
        assert sorted(o1.tolist()) == list(range(first, first + count))
        assert np.all(np.diff(key[o1]) <= 0)                       # longest first
        assert np.array_equal(o4[:count], o1)                      # the four-per-wave lookup is the same order ...
        assert np.all(o4[count:] == o1[-1])                        # ... spare slots repeat the last position
        # inside a bucket: the order of arrival
        for k in np.unique(key[o1]):
            assert [e for e in o1 if key[e] == k] == [e for e in arrival if key[e] == k]
    # no tickets yet: the stored order of the part, or the identity
    order = np.arange(n, dtype=np.int32); order[64:195] = 64 + rng.permutation(131)
    o1, o4 = emu.dispatch(n, 64, 131, np.zeros(n), np.zeros(n), np.arange(64, 195), order=order, with_tickets=False)
    assert np.array_equal(o1, order[64:195]) and np.array_equal(o4[:131], o1)
    o1, _ = emu.dispatch(n, 64, 131, np.zeros(n), np.zeros(n), np.arange(64, 195), with_tickets=False)
    assert np.array_equal(o1, np.arange(64, 195))
    # tickets that do not cover the part exactly (an env missing) are not trusted: identity
    o1, _ = emu.dispatch(n, 0, 50, rng.randint(0, 30, n), np.zeros(n), np.arange(49))
    assert np.array_equal(o1, np.arange(50))


def _heavy_states(lo, hi, want, seeds=(13, 5, 7)):
    out = []
    for sd in seeds:
        try:
            i, q, v = H.many_row_states(lo, hi, want=want, seed=sd)
        except AssertionError:
            continue
        out += [(i[k], q[k], v[k]) for k in range(len(i))]
    return out


def test_packed_third_row_set_33_to_40_rows_stays_on_the_packed_path():
    """slot_kernel.h slot_constraint<3>: environments with 33 .. 40 constraint rows (a standing humanoid: 32 contact rows + joint limits) are solved by
    the packed path itself — two full row sets plus the partial third one whose residuals are formed from the forces once per sweep — instead of
    being handed to the one-env code.  The three-set code lives in its own instantiation of the step (MAXR = SLOT_MAXROWS), which a horizon launch's
    wave calls while one of its environments is within three rows of the two-set capacity (slot_step.h slot_rollout); per-step launches keep the lean
    instantiation and their redo list.  On the fibre testbench, through the horizon form: states with 34 .. 40 rows share waves with lighter ones; every
    env agrees with the oracle step by step, nothing is re-stepped (redo total 0), and an environment with <= 32 rows gets bit-identical results
    whether or not a heavier one shares its wave (its surplus terms are exact zeros; its wave runs the other instantiation of the same arithmetic)."""
    from tests.emu.emu import EmuBatch
    from oracle import oracle as O
    mc = H.mocap()
    heavy = [h for h in _heavy_states(32, 40, want=4)]
    assert len(heavy) >= 3
    n, T = 8, 3
    idx, q, v, _ws, _c = H.varied_states(n, seed=21)
    light = (idx.copy(), q.copy(), v.copy())
    slots = [1, 4, 6, 7][:len(heavy)]                    # wave 0: one heavy + three light; wave 1: up to three heavy + one light
    for sl_, h in zip(slots, heavy):
        idx[sl_], q[sl_], v[sl_] = h
    acts = np.random.RandomState(6).randn(T, n, 28) * 0.9

    def run(states, horizon=True, mode=1):
        b = EmuBatch(H.compiled_model(), mc.data_config, mc.data_vel, n, 0)
        b.set_option(A.OPT_PACKED, mode)
        b.set(A.F_QACC_WARMSTART, np.zeros((n, 34))); b.set(A.F_TIME, np.zeros(n))
        b.set_state(states[1], states[2], frame_idx=states[0])
        ne0 = b.get(A.F_NEFC).copy()
        if horizon:
            obs, rew, done = b.rollout(acts)
        else:
            obs = np.zeros((T, n, 56)); rew = np.zeros((T, n)); done = np.zeros((T, n), dtype=np.uint8)
            for t in range(T):
                b.step(acts[t], 1, out=(obs[t], rew[t], done[t]))
        return obs, done, ne0, b.get(A.F_QPOS), b.get(A.F_QACC_WARMSTART), b.get(A.F_NEFC), b.get(A.F_SOLVER_ITER), b.redo_total()

    obs, done, ne0, qf, wsf, nef, itf, redo = run((idx, q, v))
    assert all(32 < ne0[s_] <= 40 for s_ in slots), ne0
    assert redo == 0, "an environment within 40 rows left the packed path"
    om = H.oracle_model()
    worst = 0.0
    for e in range(n):
        od = O.Data(om); od.reset(); od.set_state(q[e], v[e])
        for t in range(T):
            o, r, d, _ = od.env_step(acts[t, e])
            worst = max(worst, H.rel_err(obs[t, e], o))
            assert bool(done[t, e]) == d
        assert int(od.get("nefc")[0]) == nef[e] and int(od.get("solver_iter")[0]) == itf[e]
    assert worst < 1e-10, worst
    # the same batch through per-step launches: the lean instantiation, heavy environments through the redo list — the same results to rounding
    obs1, done1, _n, qf1, _w, nef1, _i, redo1 = run((idx, q, v), horizon=False)
    assert redo1 >= len(slots) and np.array_equal(nef1, nef) and np.array_equal(done1, done) and H.rel_err(obs1, obs) < 1e-10
    # ... and through the per-step launches with the three-set code (OPT_PACKED 2, k_step_packed_ext): nothing re-stepped, the horizon form's bits
    obs3, done3, _n, qf3, wsf3, nef3, itf3, redo3 = run((idx, q, v), horizon=False, mode=2)
    assert redo3 == 0 and np.array_equal(obs3, obs) and np.array_equal(qf3, qf) and np.array_equal(wsf3, wsf) and np.array_equal(itf3, itf)
    # the light environments next to a heavy one (three-set instantiation) and among themselves (lean instantiation): the same bits
    obs2, _d, ne2, qf2, wsf2, nef2, itf2, _r = run(light)
    same = [e for e in range(n) if e not in slots]
    for e in same:
        assert np.array_equal(obs[:, e], obs2[:, e]), e
        assert np.array_equal(qf[e], qf2[e]) and np.array_equal(wsf[e], wsf2[e]) and itf[e] == itf2[e]
