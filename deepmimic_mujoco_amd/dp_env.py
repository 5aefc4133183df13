"""`DPEnv` / `DPVecEnv`: the reference's environment API on top of the HIP batch.

`DPEnv` keeps the surface `trpo.py` consumes from `dp_env_v3.DPEnv` (src/dp_env_v3.py:34-171; callers at
src/trpo.py:30-32,66,78-79,343,403-404,461-463): `reset() -> ob[56]`, `step(ac[28]) -> (ob, reward, done, {})`,
`seed`, `close`, `action_space`, `observation_space`, `reset_model`, `reset_model_init`, `load_mocap`,
`calc_config_reward`, `is_done`, `goto`, `get_time`, `set_state`, and the attributes `sim.data.qpos/qvel/xipos/ctrl/
time`, `model.{nq,nv,nu,body_mass,opt.timestep}`, `mocap`, `idx_curr`, `idx_init`, `mocap_dt`, `mocap_data_len`,
`init_qpos`, `init_qvel`, `np_random`.  It is one environment of a `Batch` of size 1; every physics call is a HIP
kernel launch (there is no CPU path — constructing it without a GPU raises).

`DPVecEnv` is the batched form (N envs in lock step, one wavefront each), following the vendored VecEnv contract
(src/utils/vec_env/__init__.py:26-100) and DummyVecEnv's auto-reset-on-done convention
(src/utils/vec_env/dummy_vec_env.py:45-56).  It accepts numpy arrays or torch CUDA tensors.

Semantics kept from the reference: reward is the constant 1.0 unless another reward mode
is selected; `frame_skip=6` is stored but `step` runs ONE mj_step; `done` is the whole-body COM height outside
[0.7, 2.0] evaluated on the derived data of the last RK4 stage; actions are not clipped in Python; the constructor
performs gym's warm-up `step(zeros)`; `reset()` = sim.reset() + reference-state-init from Python's global `random`;
`reset_model_init()` perturbs the init pose with `self.np_random` and keeps time / warm-start.
"""
import math
import random

import numpy as np

from . import _abi as A
from .batch import Batch
from .config import Config
from .humanoid import humanoid_spec
from .mjcf import load_mjcf
from .mocap import MocapDM
from .model import CompiledModel
from .spaces import Box

REWARD_MODES = {"alive": 0, "v3-config": 1, "v2-pose": 2, "imitation": 3, "v1-quat": 4}


def _load_model(xml_path=None, explicit=True):
    """An explicit `xml_path` must exist; `Config.xml_path` (the reference's cwd-relative default, src/config.py:16) falls back
    to the built-in table of the same model (humanoid.py) when the reference tree is not the working directory."""
    import os
    if xml_path and os.path.isfile(xml_path):
        return CompiledModel(load_mjcf(xml_path))
    if xml_path and explicit:
        raise FileNotFoundError("model file %r does not exist" % (xml_path,))
    return CompiledModel(humanoid_spec())


class _Opt(object):
    def __init__(self, timestep):
        self.timestep = timestep


class _ModelView(object):
    """The handful of `sim.model` attributes the reference's scripts read."""

    def __init__(self, cm):
        self.nq, self.nv, self.nu, self.nbody = cm.nq, cm.nv, cm.nu, cm.nbody
        self.body_mass = cm.body_mass.copy()
        self.actuator_ctrlrange = cm.actuator_ctrlrange.copy()
        self.opt = _Opt(cm.timestep)


class _DataView(object):
    """`sim.data` look-alike for env 0 of a batch: attribute reads fetch from the device (copies)."""

    def __init__(self, batch, env=0):
        self._b, self._e = batch, env

    qpos = property(lambda self: self._b.get(A.F_QPOS)[self._e])
    qvel = property(lambda self: self._b.get(A.F_QVEL)[self._e])
    xipos = property(lambda self: self._b.get(A.F_XIPOS)[self._e])
    ctrl = property(lambda self: self._b.get(A.F_CTRL)[self._e])
    time = property(lambda self: float(self._b.get(A.F_TIME)[self._e]))
    qacc_warmstart = property(lambda self: self._b.get(A.F_QACC_WARMSTART)[self._e])
    ncon = property(lambda self: int(self._b.get(A.F_NCON)[self._e]))


class _SimView(object):
    def __init__(self, env):
        self._env = env
        self.data = _DataView(env._batch)
        self.model = env.model

    def forward(self):   # sim.forward(): recompute derived quantities from the current state
        b = self._env._batch
        b.set_state(b.get(A.F_QPOS), b.get(A.F_QVEL))

    def step(self):      # sim.step() with the stored ctrl
        b = self._env._batch
        b.step(b.get(A.F_CTRL))

    def reset(self):     # mj_resetData
        self._env._batch.reset(mode=2, hard=1)


class DPEnv(object):
    metadata = {"render.modes": []}
    reward_range = (-float("inf"), float("inf"))
    spec = None

    def __init__(self, motion=None, mocap_path=None, xml_path=None, device=0, reward="alive", batch_factory=None):
        self.mocap = MocapDM()
        self._cm = _load_model(xml_path if xml_path is not None else Config.xml_path, explicit=xml_path is not None)
        self.model = _ModelView(self._cm)
        self._device = device
        self._batch_factory = batch_factory or (lambda cm, cfg, vel, n, dt: Batch(cm, cfg, vel, n, device=device, mocap_dt=dt))
        self._batch = None
        self._reward_mode = REWARD_MODES[reward]
        # reward weights / scales (src/dp_env_v3.py:42-53; unused by the default reward)
        self.weight_pose, self.weight_vel, self.weight_root, self.weight_end_eff, self.weight_com = 0.5, 0.05, 0.2, 0.15, 0.1
        self.scale_pose, self.scale_vel, self.scale_end_eff, self.scale_root, self.scale_com, self.scale_err = 2.0, 0.1, 40.0, 5.0, 10.0, 1.0
        if mocap_path is None:
            mocap_path = motion if motion is not None else Config.mocap_path
        self.load_mocap(mocap_path)
        self.reference_state_init()
        self.idx_curr = -1
        self.idx_tmp_count = -1
        # --- gym MujocoEnv.__init__(xml, 6) ---------------------------------------------------------------
        self.frame_skip = 6
        self.init_qpos = self._cm.qpos0.copy()
        self.init_qvel = np.zeros(self._cm.nv)
        self.sim = _SimView(self)
        self.data = self.sim.data
        observation, _r, done, _i = self.step(np.zeros(self._cm.nu))   # the base class's warm-up step
        assert not done
        cr = self._cm.actuator_ctrlrange
        self.action_space = Box(low=cr[:, 0], high=cr[:, 1], dtype=np.float32)
        self.observation_space = Box(low=-np.inf, high=np.inf, shape=(observation.size,), dtype=np.float32)
        self.seed()

    # ---- plumbing ----------------------------------------------------------------------------------------
    @property
    def unwrapped(self):
        return self

    @property
    def dt(self):
        return self._cm.timestep * self.frame_skip

    def seed(self, seed=None):
        self.np_random = np.random.RandomState(seed)
        return [seed]

    def close(self):
        if self._batch is not None:
            self._batch.close(); self._batch = None

    def render(self, mode="human"):
        raise NotImplementedError("rendering is outside the accelerated path")

    def viewer_setup(self):
        pass

    def _sync_frame_idx(self):
        self._batch.set(A.F_FRAME_IDX, np.array([max(self.idx_curr, 0)], dtype=np.int32))
        self._batch.set(A.F_FRAME_INIT, np.array([self.idx_init], dtype=np.int32))

    # ---- reference API -------------------------------------------------------------------------------------
    def load_mocap(self, filepath):
        self.mocap.load_mocap(filepath)
        self.mocap_dt = self.mocap.dt
        self.mocap_data_len = len(self.mocap.data)
        if self._batch is not None:
            self._batch.close()
        self._batch = self._batch_factory(self._cm, self.mocap.data_config, self.mocap.data_vel, 1, float(self.mocap_dt))
        self._batch.set_option(A.OPT_REWARD_MODE, self._reward_mode)
        if hasattr(self, "sim"):
            self.sim = _SimView(self); self.data = self.sim.data

    def _get_obs(self):
        return self._batch.get_obs()[0].copy()

    def reference_state_init(self):
        self.idx_init = random.randint(0, self.mocap_data_len - 1)
        # dp_env_v3 starts its frame cursor at the draw (src/dp_env_v3.py:67-71); dp_env_v2 counts steps from 0 and adds idx_init
        # when it looks the target frame up (src/dp_env_v2.py:68-70,128-129)
        self.idx_curr = 0 if self._reward_mode in (REWARD_MODES["v2-pose"], REWARD_MODES["v1-quat"]) else self.idx_init
        self.idx_tmp_count = 0

    def early_termination(self):
        pass

    def get_joint_configs(self):
        return self.sim.data.qpos[7:]

    def calc_config_errs(self, env_config, mocap_config):
        assert len(env_config) == len(mocap_config)
        return np.sum(np.abs(env_config - mocap_config))

    def calc_config_reward(self):
        assert len(self.mocap.data) != 0
        target_config = self.mocap.data_config[self.idx_curr][7:]
        self.curr_frame = target_config
        err_configs = self.calc_config_errs(self.get_joint_configs(), target_config)
        reward_config = math.exp(-err_configs)
        self.idx_curr += 1
        self.idx_curr = self.idx_curr % self.mocap_data_len
        return reward_config

    def do_simulation(self, ctrl, n_frames):
        a = np.ascontiguousarray(np.asarray(ctrl, dtype=np.float64).reshape(1, self._cm.nu))
        return self._batch.step(a, n_substeps=int(n_frames))

    def step(self, action):
        self.step_len = 1
        if self._reward_mode != 0:
            self._sync_frame_idx()
        obs, rew, done = self.do_simulation(action, 1)
        if self._reward_mode != 0:
            self.idx_curr = int(self._batch.get(A.F_FRAME_IDX)[0])
        return obs[0].copy(), float(rew[0]), bool(done[0]), dict()

    def is_done(self):
        z_com = float(self._batch.get(A.F_COM_Z)[0])
        return bool((z_com < 0.7) or (z_com > 2.0))

    def goto(self, pos):
        self._batch.set_state(np.asarray(pos, dtype=np.float64).reshape(1, -1), self._batch.get(A.F_QVEL))

    def get_time(self):
        return self.sim.data.time

    def set_state(self, qpos, qvel):
        qpos = np.asarray(qpos, dtype=np.float64); qvel = np.asarray(qvel, dtype=np.float64)
        assert qpos.shape == (self._cm.nq,) and qvel.shape == (self._cm.nv,)
        self._batch.set_state(qpos.reshape(1, -1), qvel.reshape(1, -1))

    def reset(self):
        self._batch.reset(mode=2, hard=1)          # sim.reset()
        return self.reset_model()

    def reset_model(self):
        self.reference_state_init()
        qpos = self.mocap.data_config[self.idx_init]
        qvel = self.mocap.data_vel[self.idx_init]
        self.set_state(qpos, qvel)
        observation = self._get_obs()
        self.idx_tmp_count = -self.step_len
        return observation

    def reset_model_init(self):
        c = 0.01
        self.set_state(self.init_qpos + self.np_random.uniform(low=-c, high=c, size=self.model.nq),
                       self.init_qvel + self.np_random.uniform(low=-c, high=c, size=self.model.nv))
        return self._get_obs()


# DPVecEnv(packed=None): batches of at least this many environments step four per wavefront.  Measured closed loop (one launch set per call, two pipelined
# sub-batches, BASELINE configs[2]; profiles/r06_ab_kernel_variants.md section 3, gpurun call a5), M env-steps/s one-env / packed kernel: 2 048 envs 7.49 / 7.33,
# 3 072: 10.59 / 10.27, 4 096: 12.41 / 13.32, 6 144: 12.39 / 18.65 — the crossover sits between 3 072 and 4 096 since round 5's cuts of the packed step
# (rounds 3-4: 12.3 / 11.4 at 4 096, hence the old threshold of 6 144).
PACKED_FROM_ENVS = 4096


class _Info(dict):
    """An env's `info` dict that tells its list when something is written into it (so that the list is not rebuilt every step)."""
    __slots__ = ("_owner",)

    def _touch(self):
        self._owner.dirty = True

    def __setitem__(self, k, v):
        self._touch(); dict.__setitem__(self, k, v)

    def update(self, *a, **kw):
        self._touch(); dict.update(self, *a, **kw)

    def setdefault(self, k, d=None):
        self._touch(); return dict.setdefault(self, k, d)

    def __ior__(self, other):
        self._touch(); dict.update(self, other); return self


class _InfoList(list):
    """The list `step_wait` hands out every step while nobody has written into it.  The reference's DummyVecEnv returns a fresh copy per step
    (src/utils/vec_env/dummy_vec_env.py:56): any write — into a dict (_Info) or into the list itself — marks it dirty, and the next step starts a new one."""

    def __init__(self, n):
        list.__init__(self, (_Info() for _ in range(n)))
        self.dirty = False
        for d in self:
            d._owner = self


def _marking(name):                                    # every list mutator marks the list before it acts
    f = getattr(list, name)

    def g(self, *a, **kw):
        self.dirty = True
        return f(self, *a, **kw)
    g.__name__ = name
    return g


_INFO_LIST_MUTATORS = ("__setitem__", "__delitem__", "__iadd__", "__imul__", "append", "extend", "insert", "pop", "remove", "clear", "sort", "reverse")
for _m in _INFO_LIST_MUTATORS:
    setattr(_InfoList, _m, _marking(_m))
del _m


class DPVecEnv(object):
    """N DeepMimic humanoids in lock step on one GPU (one wavefront per environment)."""

    def __init__(self, num_envs, motion="walk", xml_path=None, device=0, reward="alive", autoreset="rsi", seed=0,
                 contacts=True, limits=True, action_mode="raw", env_offset=0, batch_factory=None, frame_skip=None, diagnostics=False, dtype=64, packed=None,
                 step_queue=0):
        """reward="imitation": the 5-term reward of code.md:1017-1143 (imitation.py) against the frame after the current one.
        frame_skip: sim steps per env step (src/dp_env_v3.py:108-112 hard-codes 1); "mocap" = floor(mocap_dt / timestep), the
        commented intent of :107-110, so that one env step spans one mocap frame.  Default (None): 1, except "mocap" for the
        imitation reward — its reference advances one mocap frame per env step and its velocity features are per second, so any
        other value plays the clip at the wrong speed (a warning says so when one is given).
        diagnostics: keep `sim.data.xipos` / the contact geom list up to date after every step (DM_OPT_DIAGNOSTICS; the batched
        training path does not read them, `DPEnv` and raw `Batch` objects default to on).
        dtype: 64 (default) or 32 — arithmetic of the kernels; observations / actions stay float64 arrays.
        packed: DM_OPT_PACKED — four environments per wavefront (k_step_packed) instead of one.  True / False pin the kernel; 2 pins the per-step launches
        with the three-set code (k_step_packed_ext: 40 constraint rows per env instead of 32, ~8 % slower otherwise — for populations that stand on both
        feet: 8.96 against 6.18 M env-steps/s one env per wave at 8 192 envs, closed loop).  None
        (default): batches of PACKED_FROM_ENVS environments or more (two or more waves per SIMD on one MI355X; float64; every reward mode, v1-quat
        included since round 6) start on the packed kernel and re-decide every 256 steps from their own row statistics (Batch.enable_auto_packed): it is
        1.4-1.5x faster while environments stay within its per-env capacities (the RSI / early-termination regimes), and hands over to
        the one-env kernel when a competent policy keeps most environments on both feet (32+ rows).  Smaller batches: one env per wave —
        except for models without contacts and limits (BASELINE configs[1]): all waves cost the same there and four per wave is 1.5x
        faster at any size.
        step_queue: DM_OPT_STEP_QUEUE depth (0 = off): queue `batch.step` calls and run them as one horizon launch (see below)."""
        self.num_envs = int(num_envs)
        self.mocap = MocapDM()
        self.mocap.load_mocap(motion)
        self.mocap_dt = self.mocap.dt
        self.mocap_data_len = len(self.mocap.data)
        self._cm = _load_model(xml_path)
        self.model = _ModelView(self._cm)
        flags = (0 if contacts else A.FLAG_NO_CONTACT) | (0 if limits else A.FLAG_NO_LIMIT)
        per_frame = max(1, int(float(self.mocap_dt) / float(self._cm.timestep)))
        if frame_skip is None:
            frame_skip = "mocap" if reward == "imitation" else 1
        self.frame_skip = per_frame if frame_skip == "mocap" else int(frame_skip)
        if reward == "imitation" and self.frame_skip != per_frame:
            import warnings
            warnings.warn("imitation reward with frame_skip=%d: the reference clip advances one frame (%.4f s) per env step of "
                          "%.4f s; use frame_skip='mocap' (= %d) to play it in sim time"
                          % (self.frame_skip, float(self.mocap_dt), self.frame_skip * float(self._cm.timestep), per_frame))
        imit = None
        if reward in ("imitation", "v1-quat"):
            from .imitation import ImitationSpec
            self.imitation = ImitationSpec(self._cm)
            imit = self.imitation.table_for(self.mocap)
        if batch_factory is None:
            self._batch = Batch(self._cm, self.mocap.data_config, self.mocap.data_vel, self.num_envs, device=device,
                                flags=flags, mocap_dt=float(self.mocap_dt), imitation=imit, dtype=dtype)
        elif imit is not None:
            self._batch = batch_factory(self._cm, self.mocap.data_config, self.mocap.data_vel, self.num_envs, flags, imitation=imit)
        else:
            self._batch = batch_factory(self._cm, self.mocap.data_config, self.mocap.data_vel, self.num_envs, flags)
        b = self._batch
        b.set_option(A.OPT_REWARD_MODE, REWARD_MODES[reward])
        b.set_option(A.OPT_AUTORESET, {"none": 0, None: 0, "rsi": 1, "init": 2}[autoreset])
        b.set_option(A.OPT_ACTION_MODE, {"raw": 0, "p-control": 1, "pd": 2}[action_mode])
        b.set_option(A.OPT_SEED, int(seed))
        b.set_option(A.OPT_ENV_OFFSET, int(env_offset))
        b.set_option(A.OPT_DIAGNOSTICS, 1 if diagnostics else 0)
        rowless = not (contacts or limits)          # no constraint rows: every wave costs the same, the packed kernel wins at any batch size
        auto = packed is None and batch_factory is None and (self.num_envs >= PACKED_FROM_ENVS or (rowless and self.num_envs >= 256)) and dtype == 64
        # a horizon launch (Batch.rollout, rollout.SegmentCollector) may use the packed kernel at ANY batch size: there a wave does not wait
        # for the slowest wave of every step
        self.horizon_packed_ok = packed is None and batch_factory is None and self.num_envs >= 256 and dtype == 64
        if packed or auto:
            b.set_option(A.OPT_PACKED, 2 if (packed is not True and packed == 2) else 1)     # (packed=2: per-step launches with the three-set code, see the docstring)
        if auto:
            b.enable_auto_packed(True)
        if step_queue:
            # DM_OPT_STEP_QUEUE (include/dmenv.h): `batch.step` calls with device tensors are queued and run together as one horizon launch when `step_queue`
            # calls are queued or at `batch.join()` / any other entry point — for open-loop callers (pre-drawn or scripted actions) that do not read a step's
            # outputs before the next call.  It rides on the packed kernels; `step()` / `step_wait()` of this class join, so the facade keeps its semantics.
            if not self.packed:
                b.set_option(A.OPT_PACKED, 1)
            b.set_option(A.OPT_STEP_QUEUE, int(step_queue))
        cr = self._cm.actuator_ctrlrange
        self.action_space = Box(low=cr[:, 0], high=cr[:, 1], dtype=np.float32)
        self.observation_space = Box(low=-np.inf, high=np.inf, shape=(A.NOBS,), dtype=np.float32)
        self._pending = None

    @property
    def batch(self):
        return self._batch

    @property
    def packed(self):
        """True while the batch steps four environments per wavefront (DM_OPT_PACKED; may change over a run with packed=None)"""
        return bool(getattr(self._batch, "options", {}).get(A.OPT_PACKED, 0))

    def seed(self, seed):
        self._batch.set_option(A.OPT_SEED, int(seed))

    def reset(self, mode="rsi", out=None):
        self._batch.reset(mode={"rsi": 0, "init": 1, "qpos0": 2}[mode], hard=1)
        return self._batch.get_obs(out)

    def step_async(self, actions):
        self._pending = actions

    def step_wait(self, out=None):
        obs, rew, done = self._batch.step(self._pending, self.frame_skip, out)
        self._pending = None
        if getattr(self._batch, "_queue_refs", None) is not None:
            self._batch.join()      # OPT_STEP_QUEUE: the call was only queued, and what this method returns is read in stream order
        # `infos`: one dict per env (src/utils/vec_env/dummy_vec_env.py:45-56).  This env never puts anything into them, so the list is
        # built once and handed out again while every dict is still empty (building 4 096 dicts per step cost more than the step's
        # launch); a caller that writes into one (bench/monitor.py:73-74 does, at episode ends) gets fresh dicts from the next step on.
        infos = self._infos
        if infos is None or infos.dirty:
            infos = self._infos = _InfoList(self.num_envs)
        return obs, rew, done, infos

    _infos = None

    def step(self, actions, out=None):
        self.step_async(actions)
        return self.step_wait(out)

    def close(self):
        self._batch.close()
