"""Host logic of DPEnv / DPVecEnv (API surface, reset/seed semantics, frame index bookkeeping) driven through the wave
testbench in place of the GPU (`batch_factory` seam); the same classes run on the real device in test_gpu_parity.py."""
import random

import numpy as np
import pytest

from deepmimic_mujoco_amd import _abi as A
from deepmimic_mujoco_amd.dp_env import DPEnv, DPVecEnv
from tests import helpers as H
from tests.emu.emu import EmuBatch


def emu_factory(cm, cfg, vel, n, dt):
    return EmuBatch(cm, cfg, vel, n, 0)


class _CloseableEmu(EmuBatch):
    def close(self):
        pass

    def get_obs(self, out=None):
        q = self.get(A.F_QPOS); v = self.get(A.F_QVEL)
        return np.concatenate([q[:, 7:], v[:, 6:]], 1)


def factory(cm, cfg, vel, n, dt):
    return _CloseableEmu(cm, cfg, vel, n, 0)


@pytest.fixture(scope="module")
def env():
    random.seed(0)
    e = DPEnv(motion="walk", batch_factory=factory)
    yield e
    e.close()


def test_gym_surface(env):
    assert env.action_space.shape == (28,) and np.all(env.action_space.low == -0.5) and np.all(env.action_space.high == 0.5)
    assert env.observation_space.shape == (56,) and np.all(np.isinf(env.observation_space.low))
    assert env.spec is None and env.unwrapped is env and env.reward_range[0] == -float("inf")
    assert env.mocap_data_len == 39 and abs(env.mocap_dt - 0.033332) < 1e-12 and env.frame_skip == 6
    assert abs(env.dt - 0.0996) < 1e-12
    assert np.array_equal(env.init_qpos, [0, 0, 0.9, 1] + [0] * 31) and env.model.nq == 35 and env.model.nu == 28
    assert env.action_space.sample().shape == (28,)


def test_constructor_runs_gyms_warmup_step():
    random.seed(3)
    e = DPEnv(motion="walk", batch_factory=factory)
    assert abs(e.get_time() - 0.0166) < 1e-15                       # one mj_step from the init pose (MujocoEnv.__init__)
    assert e.sim.data.qpos[2] < 0.9 and np.abs(e.sim.data.qacc_warmstart).max() > 0
    e.close()


def test_reset_is_sim_reset_plus_rsi(env):
    random.seed(11)
    expect = random.randint(0, 38)
    random.seed(11)
    ob = env.reset()
    assert env.idx_init == expect == env.idx_curr and env.idx_tmp_count == -1
    assert np.array_equal(ob, np.concatenate([env.mocap.data_config[expect][7:], env.mocap.data_vel[expect][6:]]))
    assert env.get_time() == 0.0 and np.all(env.sim.data.qacc_warmstart == 0)
    assert np.array_equal(env.sim.data.qpos, env.mocap.data_config[expect])


def test_step_returns_reference_tuple(env):
    random.seed(5)
    env.reset()
    ob, r, d, info = env.step(np.zeros(28))
    assert ob.shape == (56,) and ob.dtype == np.float64 and r == 1.0 and d is False and info == {}
    assert np.array_equal(ob[:28], env.sim.data.qpos[7:]) and np.array_equal(ob[28:], env.sim.data.qvel[6:])
    assert abs(env.get_time() - 0.0166) < 1e-15
    assert env.is_done() is False


def test_reset_model_init_semantics(env):
    env.seed(7)
    rs = np.random.RandomState(7)
    random.seed(5); env.reset(); env.step(np.zeros(28))
    t_before = env.get_time()
    ob = env.reset_model_init()
    q = env.init_qpos + rs.uniform(low=-0.01, high=0.01, size=35)
    v = env.init_qvel + rs.uniform(low=-0.01, high=0.01, size=34)
    assert np.array_equal(env.sim.data.qpos, q) and np.array_equal(env.sim.data.qvel, v)      # quaternion NOT renormalised
    assert np.array_equal(ob, np.concatenate([q[7:], v[6:]]))
    assert env.get_time() == t_before                                                          # set_state keeps time


def test_calc_config_reward_and_index_wrap(env):
    random.seed(1); env.reset()
    env.idx_curr = 38
    q = env.sim.data.qpos
    r = env.calc_config_reward()
    assert abs(r - np.exp(-np.abs(q[7:] - env.mocap.data_config[38][7:]).sum())) < 1e-15 and env.idx_curr == 0


def test_set_state_and_goto(env):
    q = env.mocap.data_config[3].copy(); v = env.mocap.data_vel[3].copy()
    env.set_state(q, v)
    assert np.array_equal(env.sim.data.qpos, q) and np.array_equal(env.sim.data.qvel, v)
    q2 = q.copy(); q2[2] += 0.1
    env.goto(q2)
    assert np.array_equal(env.sim.data.qpos, q2) and abs(env.sim.data.xipos[1][2] - (q2[2] + 0.07 * 1.0)) < 0.08
    with pytest.raises(AssertionError):
        env.set_state(q[:-1], v)


def test_v3_config_reward_mode_advances_frame_index():
    random.seed(2)
    e = DPEnv(motion="walk", reward="v3-config", batch_factory=factory)
    random.seed(2); e.reset()
    i0 = e.idx_curr
    q_before = None
    ob, r, d, _ = e.step(np.zeros(28))
    assert e.idx_curr == (i0 + 1) % e.mocap_data_len
    assert abs(r - np.exp(-np.abs(e.sim.data.qpos[7:] - e.mocap.data_config[i0][7:]).sum())) < 1e-14
    e.close()


def test_vec_env_surface():
    def vf(cm, cfg, vel, n, flags):
        return _CloseableEmu(cm, cfg, vel, n, flags)
    env = DPVecEnv(3, motion="walk", batch_factory=vf, autoreset="rsi", seed=4)
    ob = env.reset("rsi")
    assert ob.shape == (3, 56) and env.num_envs == 3
    obs, rew, done, infos = env.step(np.zeros((3, 28)))
    assert obs.shape == (3, 56) and rew.shape == (3,) and done.shape == (3,) and np.all(rew == 1.0)
    env.close()


def _vf(cm, cfg, vel, n, flags, imitation=None):
    return _CloseableEmu(cm, cfg, vel, n, flags, imitation=imitation)


@pytest.mark.parametrize("reward", ["v3-config", "imitation"])
def test_autoreset_init_redraws_the_frame_like_env_reset_then_reset_model_init(reward):
    """The trainer's episode start (src/trpo.py:78-79) is env.reset() — sim.reset() + RSI, which draws idx_init / idx_curr
    (src/dp_env_v3.py:67-71,148-156) — followed by reset_model_init(), which overrides the state only.  Autoreset "init" must
    therefore leave a freshly drawn frame (and cycle 0) behind, and the frame-indexed rewards of the new episode must follow
    it: checked against the oracle fed with the frame the device's counter-based stream predicts."""
    from oracle import oracle as O
    n, seed, off = 3, 9, 5
    env = DPVecEnv(n, motion="walk", batch_factory=_vf, reward=reward, autoreset="init", seed=seed, env_offset=off, frame_skip=1)
    b = env.batch
    mc = env.mocap; F = len(mc.data_config)
    env.reset("rsi")
    ep = b.get(A.F_EPISODE).copy()
    q = b.get(A.F_QPOS); q[:, 2] = 0.5; q[1, 2] = mc.data_config[0][2]            # envs 0 and 2 start below the termination height
    b.set_state(q, b.get(A.F_QVEL))
    b.set(A.F_FRAME_IDX, np.array([F - 1, 7, 11], dtype=np.int32)); b.set(A.F_CYCLE, np.array([2, 0, 1], dtype=np.int32))
    obs, rew, done, _ = env.step(np.zeros((n, 28)))
    assert list(done.astype(bool)) == [True, False, True]
    fi = b.get(A.F_FRAME_IDX); fin = b.get(A.F_FRAME_INIT); cyc = b.get(A.F_CYCLE)
    for e in (0, 2):
        want = H.device_rsi_frame(seed, off + e, int(ep[e]), F)
        assert fi[e] == want == fin[e] and cyc[e] == 0, (e, fi[e], want)
        assert np.all(np.abs(b.get(A.F_QPOS)[e] - env._cm.qpos0) <= 0.01 + 1e-15)      # state: noisy init pose, not the mocap frame
    assert fi[1] == 8 and cyc[1] == 0
    # the next step's reward is computed against the redrawn frame: oracle with that frame index
    om = H.oracle_model(); od = O.Data(om)
    qn = b.get(A.F_QPOS); vn = b.get(A.F_QVEL)
    a = np.zeros((n, 28))
    obs2, rew2, done2, _ = env.step(a)
    for e in (0, 2):
        od.reset(); od.set_state(qn[e], vn[e])
        if reward == "imitation":
            T, P = env.imitation.table_for(mc)
            o, r, d, ic, cy = O.env_step_imitation(om, od, a[e], 1, T, P, int(fi[e]), 0)
        else:
            o, r, d, ic = od.env_step(a[e], 1, 1, mc.data_config, int(fi[e]), int(fin[e]))
        assert abs(r - rew2[e]) < 1e-10 and H.rel_err(obs2[e], o) < 1e-10 and b.get(A.F_FRAME_IDX)[e] == ic
    env.close()


def test_v2_pose_cursor_counts_from_zero_after_reset_and_set_state():
    """dp_env_v2.reference_state_init keeps the draw in idx_init and sets idx_curr = 0; the target frame is
    (idx_curr + idx_init) % F (src/dp_env_v2.py:68-70,128-129) — after a device reset or a frame-indexed set_state as well."""
    from oracle import oracle as O
    n = 2
    env = DPVecEnv(n, motion="walk", batch_factory=_vf, reward="v2-pose", autoreset="rsi", seed=3)
    b = env.batch; mc = env.mocap; F = len(mc.data_config)
    env.reset("rsi")
    fin = b.get(A.F_FRAME_INIT)
    assert np.all(b.get(A.F_FRAME_IDX) == 0) and fin[0] == H.device_rsi_frame(3, 0, 0, F)
    b.set_state(mc.data_config[[4, 30]], mc.data_vel[[4, 30]], frame_idx=np.array([4, 30], dtype=np.int32))
    assert np.all(b.get(A.F_FRAME_IDX) == 0) and list(b.get(A.F_FRAME_INIT)) == [4, 30]
    om = H.oracle_model(); a = np.zeros((n, 28))
    obs, rew, done, _ = env.step(a)
    for e, k in enumerate((4, 30)):
        od = O.Data(om); od.reset(); od.set_state(mc.data_config[k], mc.data_vel[k])
        o, r, d, ic = od.env_step(a[e], 1, 2, mc.data_config, 0, k)                 # target frame (0 + 1 + k) % F
        assert abs(r - rew[e]) < 1e-12 and ic == 1 == b.get(A.F_FRAME_IDX)[e]
    random.seed(6)
    e1 = DPEnv(motion="walk", reward="v2-pose", batch_factory=factory)
    random.seed(6); e1.reset()
    assert e1.idx_curr == 0 and 0 <= e1.idx_init < F
    e1.step(np.zeros(28))
    assert e1.idx_curr == 1
    e1.close(); env.close()


def test_defaults_frame_skip_follows_the_reward_and_a_missing_model_file_is_an_error(tmp_path):
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        e = DPVecEnv(1, motion="walk", batch_factory=_vf, reward="imitation")
        assert e.frame_skip == 2                                                      # floor(0.0333 / 0.0166)
        e2 = DPVecEnv(1, motion="walk", batch_factory=_vf, reward="alive")
        assert e2.frame_skip == 1
    with pytest.warns(UserWarning, match="frame_skip"):
        DPVecEnv(1, motion="walk", batch_factory=_vf, reward="imitation", frame_skip=1)
    with pytest.raises(FileNotFoundError):
        DPVecEnv(1, motion="walk", batch_factory=_vf, xml_path=str(tmp_path / "nope.xml"))
    with pytest.raises(FileNotFoundError):
        DPEnv(motion="walk", xml_path=str(tmp_path / "nope.xml"), batch_factory=factory)


def test_bare_dpenv_uses_the_committed_default_clip():
    """`DPEnv()` = the reference's committed configuration, Config.motion == 'dance_b' (src/config.py:9): 153 frames at 60 Hz."""
    from deepmimic_mujoco_amd.config import Config
    assert Config.motion == "dance_b"
    random.seed(4)
    e = DPEnv(batch_factory=factory)
    mc = H.mocap("dance_b")
    assert e.mocap_data_len == 153 and abs(e.mocap_dt - mc.dt) < 1e-15
    random.seed(9); expect = random.randint(0, 152); random.seed(9)
    ob = e.reset()
    assert e.idx_init == expect and np.array_equal(ob, np.concatenate([mc.data_config[expect][7:], mc.data_vel[expect][6:]]))
    ob, r, d, info = e.step(np.zeros(28))
    assert r == 1.0 and d is False and np.isfinite(ob).all()
    e.close()


def test_vec_env_step_wait_host_overhead_is_small_at_4096_envs():
    """`DPVecEnv.step` is the drop-in for `VecEnv.step` (src/utils/vec_env/__init__.py:26-100, dummy_vec_env.py:45-56): what it adds on the
    host to `Batch.step` must stay far below a step's 250-350 us at 4 096 envs (building 4 096 fresh info dicts per step took 220 us)."""
    import time

    class _NullBatch(object):           # a batch whose step costs nothing: what is timed is the facade
        def __init__(self, n):
            self.n = n; self.out = (np.zeros((n, 56)), np.zeros(n), np.zeros(n, dtype=np.uint8))

        def set_option(self, *a):
            pass

        def step(self, action, n_substeps=1, out=None):
            return self.out

    n = 4096
    env = DPVecEnv(n, motion="walk", batch_factory=lambda cm, cfg, vel, nn, flags: _NullBatch(nn), autoreset="rsi")
    act = np.zeros((n, 28))
    null = env.batch
    for _ in range(50):
        env.step(act)
    reps = 2000
    t0 = time.perf_counter()
    for _ in range(reps):
        null.step(act, 1, None)
    base = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        env.step(act)
    per = (time.perf_counter() - t0) / reps - base
    assert per < 20e-6, "DPVecEnv.step adds %.1f us per step on the host" % (per * 1e6)
    # the contract of the infos: one dict per env, empty; a caller's write is not seen by later steps
    obs, rew, done, infos = env.step(act)
    assert len(infos) == n and all(isinstance(d, dict) and not d for d in infos)
    infos[7]["episode"] = {"r": 1.0}
    infos2 = env.step(act)[3]
    assert infos2 is not infos and not infos2[7] and infos[7] == {"episode": {"r": 1.0}}
    infos3 = env.step(act)[3]
    assert infos3 is infos2
    for write in (lambda d: d.update(a=1), lambda d: d.setdefault("a", 1)):
        cur = env.step(act)[3]
        write(cur[0])
        assert env.step(act)[3] is not cur
    # ... nor is a write into the LIST (the reference hands out a copy per step: src/utils/vec_env/dummy_vec_env.py:56)
    for write in (lambda l: l.__setitem__(3, {"x": 1}), lambda l: l.append({}), lambda l: l.__delitem__(0), lambda l: l.extend([{}]), lambda l: l.pop()):
        cur = env.step(act)[3]
        write(cur)
        nxt = env.step(act)[3]
        assert nxt is not cur and len(nxt) == n and all(not d for d in nxt)
    # every listed mutator flips .dirty; copies and slices are plain lists (the caller's own objects, nothing to track)
    from deepmimic_mujoco_amd.dp_env import _InfoList, _INFO_LIST_MUTATORS
    args = {"__setitem__": (0, {}), "__delitem__": (0,), "__iadd__": ([{}],), "__imul__": (1,), "append": ({},), "extend": ([{}],), "insert": (0, {}), "pop": (), "remove": None,
            "clear": (), "sort": None, "reverse": ()}
    assert set(args) == set(_INFO_LIST_MUTATORS)
    for name in _INFO_LIST_MUTATORS:
        l = _InfoList(3)
        assert not l.dirty and getattr(_InfoList, name) is not getattr(list, name)
        a = args[name]
        if name == "remove":
            a = (l[0],)
        if name == "sort":
            getattr(l, name)(key=id)
        else:
            getattr(l, name)(*a)
        assert l.dirty, name
    l = _InfoList(3)
    l[0:2] = [{}, {}]
    assert l.dirty
    l = _InfoList(3)
    assert type(l.copy()) is list and type(l[0:2]) is list and not l.dirty


def test_per_step_kernel_chooser_is_a_function_of_the_redo_statistics():
    """`Batch._adapt` (host logic, no device): lean packed -> three-set per-step kernel when the redo rate is high and nearly all of it row overflows;
    -> one-env kernel for overflows of another kind, or when the three-set kernel still overflows; three-set -> lean when nobody holds more than 32 rows;
    one-env -> lean / three-set by the largest row count."""
    import numpy as np
    from deepmimic_mujoco_amd import _abi as A
    from deepmimic_mujoco_amd.batch import Batch

    class Fake(Batch):
        def __init__(self):
            self.n = 100; self.options = {A.OPT_PACKED: 1}; self.reasons = [0, 0, 0, 0, 0, 0]; self.nefc = np.zeros(100, dtype=np.int32)
            self._auto = True; self._auto_ctr = 0; self._redo_last = 0; self._redo_rows_last = 0; self.auto_switches = 0; self._calm_checks = 0
            self.ADAPT_EVERY = 1

        def set_option(self, o, v):
            self.options[int(o)] = int(v)

        def redo_reasons(self):
            return list(self.reasons)

        def redo_total(self):
            return self.reasons[0]

        def get(self, f):
            assert f == A.F_NEFC
            return self.nefc

        def __del__(self):
            pass
    b = Fake()
    b._adapt(); assert b.options[A.OPT_PACKED] == 1 and b.auto_switches == 0                # nothing re-stepped: stays
    b.reasons = [10, 0, 0, 0, 10, 0]                                                          # 10 % of env-steps, all for rows: the three-set kernel
    b._adapt(); assert b.options[A.OPT_PACKED] == 2 and b.auto_switches == 1
    b.nefc[:] = 20; b.nefc[3] = 35
    b._adapt(); assert b.options[A.OPT_PACKED] == 2 and b.auto_switches == 1                # no new overflows, somebody above 32 rows: stays
    b.nefc[3] = 30
    # nobody above 32 — but one snapshot is not evidence (round 6: a population near the redo threshold reads zero one look in twelve): three looks in a row
    b._adapt(); assert b.options[A.OPT_PACKED] == 2 and b.auto_switches == 1
    b._adapt(); assert b.options[A.OPT_PACKED] == 2 and b.auto_switches == 1
    b.nefc[3] = 33
    b._adapt(); assert b.options[A.OPT_PACKED] == 2 and b._calm_checks == 0                  # somebody is back above 32: the count starts again
    b.nefc[3] = 30
    for _ in range(Batch.CALM_CHECKS - 1):
        b._adapt(); assert b.options[A.OPT_PACKED] == 2 and b.auto_switches == 1
    b._adapt(); assert b.options[A.OPT_PACKED] == 1 and b.auto_switches == 2                # the lean kernel again
    b.reasons = [20, 0, 0, 9, 11, 0]                                                          # overflows of another kind (contacts): the one-env kernel
    b._adapt(); assert b.options[A.OPT_PACKED] == 0 and b.auto_switches == 3
    b.nefc[3] = 45
    b._adapt(); assert b.options[A.OPT_PACKED] == 0                                          # an environment beyond every packed capacity: stays
    b.nefc[3] = 36
    b._adapt(); assert b.options[A.OPT_PACKED] == 2 and b.auto_switches == 4                # within 38 rows: the three-set kernel
    b.reasons = [40, 0, 0, 0, 31, 0]                                                          # it overflows all the same: the one-env kernel
    b._adapt(); assert b.options[A.OPT_PACKED] == 0 and b.auto_switches == 5
    b.nefc[:] = 12
    b._adapt(); assert b.options[A.OPT_PACKED] == 1 and b.auto_switches == 6
    # the collector's horizon launches count their in-wave re-steps into the same counters while the chooser is suspended: its restore() re-baselines
    b.reasons = [5000, 0, 0, 0, 5000, 0]
    b.rebaseline_auto()
    b._adapt(); assert b.options[A.OPT_PACKED] == 1 and b.auto_switches == 6                # nothing of that is this window's evidence
