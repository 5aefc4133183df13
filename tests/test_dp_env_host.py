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
