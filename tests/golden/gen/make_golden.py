#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by EXECUTING the reference's own Python.

Runs only inside the build container (needs /root/reference).  Nothing from the reference is copied:
the reference modules are imported from where they lie, driven with seeded inputs, and only
input/output DATA is written out.

What is "true golden" (reference code end to end):
  * euler_from_quaternion(.,'rxyz')            src/transformations.py:1089-1097 (+1031-1086,1174-1193)
  * DPEnv._get_obs / calc_config_errs / calc_config_reward / is_done / reference_state_init
                                                 src/dp_env_v3.py:62-71,85-104,134-139
  * dp_env_v2 calc_config_errs / calc_vel_errs / calc_reward    src/dp_env_v2.py:91-172
  * MujocoInterface kp/kd tables, calc_root_errs, calc_vel_errs  src/mujoco/mujoco_interface.py:66-72,192-210
What is "semi golden" (reference code, but the quaternion primitive is the stand-in in
stubs/pyquaternion because pyquaternion is not installed here):
  * MocapDM.load_mocap -> data / data_config / data_vel            src/mujoco/mocap_v2.py:20-149
  * align_rotation / align_position / calc_diff_from_quaternion / calc_angular_vel_from_quaternion
                                                 src/mujoco/mocap_util.py:31-77
  * MujocoInterface.calc_config_errs / calc_config_err_vec         src/mujoco/mujoco_interface.py:119-190
  * dp_env_v2.calc_root_errs                                       src/dp_env_v2.py:101-114

Also writes deepmimic_mujoco_amd/assets/motions.npz: the raw mocap `Frames` arrays (input DATA of the
clips under src/mujoco/motions/), repacked as float64 arrays so the product can run where
/root/reference does not exist (the GPU box).
"""
import json
import os
import random
import sys
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", "..", ".."))
REF = "/root/reference/src"
OUT = os.path.join(REPO, "tests", "golden")
ASSETS = os.path.join(REPO, "deepmimic_mujoco_amd", "assets")

CLIPS = ["backflip", "cartwheel", "crawl", "dance_a", "dance_b", "getup_facedown", "getup_faceup",
         "jump", "kick", "punch", "roll", "run", "spin", "spinkick", "walk"]


def setup_imports():
    sys.path.insert(0, os.path.join(HERE, "stubs"))
    sys.path.insert(0, REF)
    os.chdir(REF)  # Config builds cwd-relative paths (src/config.py:7-17)
    warnings.simplefilter("ignore")


class FakeData(object):
    def __init__(self, qpos, qvel, xipos, ctrl=None):
        self.qpos = np.array(qpos, dtype=np.float64)
        self.qvel = np.array(qvel, dtype=np.float64)
        self.xipos = np.array(xipos, dtype=np.float64)
        self.ctrl = np.zeros(28) if ctrl is None else np.array(ctrl, dtype=np.float64)
        self.time = 0.0


class FakeSim(object):
    def __init__(self, data):
        self.data = data


class FakeModel(object):
    def __init__(self, body_mass):
        self.body_mass = np.array(body_mass, dtype=np.float64)


def gen_mocap():
    from mujoco.mocap_v2 import MocapDM
    raw = {}
    for clip in CLIPS:
        path = os.path.join(REF, "mujoco", "motions", "humanoid3d_%s.txt" % clip)
        with open(path) as f:
            js = json.load(f)
        frames = np.array(js["Frames"], dtype=np.float64)
        raw["frames_" + clip] = frames
        raw["loop_" + clip] = np.array(js.get("Loop", "none"))
        m = MocapDM()
        m.load_mocap(path)
        np.savez(os.path.join(OUT, "mocap_%s.npz" % clip),
                 dt=np.float64(m.dt), durations=np.array(m.durations, dtype=np.float64),
                 data=np.array(m.data, dtype=np.float64),
                 data_config=np.array(m.data_config, dtype=np.float64),
                 data_vel=np.array(m.data_vel, dtype=np.float64))
        print("mocap", clip, np.array(m.data).shape, "max|vel| %.3f" % np.abs(np.array(m.data_vel)).max())
    os.makedirs(ASSETS, exist_ok=True)
    np.savez_compressed(os.path.join(ASSETS, "motions.npz"), **raw)


def gen_euler():
    import transformations as T
    rng = np.random.RandomState(1234)
    q = rng.randn(1000, 4)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    # edge cases: identity, exact +-90 deg about y (gimbal branch cy<=_EPS), tiny rotations, unnormalised
    extra = [[0, 0, 0, 1], [0, np.sqrt(0.5), 0, np.sqrt(0.5)], [0, -np.sqrt(0.5), 0, np.sqrt(0.5)],
             [1e-9, 0, 0, 1], [0, 0, 1e-9, 1], [0.5, 0.5, 0.5, 0.5], [2.0, 0, 0, 2.0],
             [0.3, np.sqrt(0.5), 0.3, np.sqrt(0.5)], [1, 0, 0, 0], [0, 0, 1, 0]]
    q = np.concatenate([q, np.array(extra, dtype=np.float64)], 0)
    out = np.array([T.euler_from_quaternion(qi, axes="rxyz") for qi in q], dtype=np.float64)
    np.savez(os.path.join(OUT, "euler_rxyz_golden.npz"), quat_xyzw=q, euler=out)
    print("euler", q.shape, out.shape)


def gen_quat_ops():
    from mujoco import mocap_util as MU
    from mujoco.mocap_v2 import MocapDM
    rng = np.random.RandomState(77)
    q0 = rng.randn(300, 4)
    q0 /= np.linalg.norm(q0, axis=1, keepdims=True)
    q1 = rng.randn(300, 4)
    q1 /= np.linalg.norm(q1, axis=1, keepdims=True)
    # close pairs (small relative rotation) and identical pairs (axis undefined -> zeros)
    q1[200:280] = q0[200:280] + 1e-3 * rng.randn(80, 4)
    q1[200:280] /= np.linalg.norm(q1[200:280], axis=1, keepdims=True)
    q1[280:] = q0[280:]
    pos = rng.randn(300, 3)
    dura = rng.uniform(0.01, 0.05, size=300)
    m = MocapDM()
    res = dict(q0=q0, q1=q1, pos=pos, dura=dura,
               align_rotation=np.array([MU.align_rotation(a) for a in q0]),
               align_position=np.array([MU.align_position(p) for p in pos]),
               calc_rot_vel=np.array([m.calc_rot_vel(a, b, d) for a, b, d in zip(q0, q1, dura)]),
               ang_vel_from_quat=np.array([MU.calc_angular_vel_from_quaternion(a, b, d)
                                           for a, b, d in zip(q0, q1, dura)]),
               diff_from_quat=np.array([MU.calc_diff_from_quaternion(a, b) for a, b in zip(q0, q1)]))
    np.savez(os.path.join(OUT, "quat_ops_golden.npz"), **res)
    print("quat ops", {k: v.shape for k, v in res.items()})


def gen_env_logic():
    """obs / done / reward / frame-index logic of dp_env_v3 (+ v2 reward, interface tables)."""
    import dp_env_v3 as E3
    from mujoco.mocap_v2 import MocapDM
    from mujoco.mujoco_interface import MujocoInterface

    walk = os.path.join(REF, "mujoco", "motions", "humanoid3d_walk.txt")
    mocap = MocapDM()
    mocap.load_mocap(walk)
    F = len(mocap.data)
    # body masses of dp_env_v3.xml (world 0 + 13 bodies; elbow = capsule 1.0 + wrist 0.5)
    body_mass = [0, 6, 14, 2, 1.5, 1.5, 1.5, 1.5, 4.5, 3, 1, 4.5, 3, 1]
    rng = np.random.RandomState(2024)
    n = 128
    qpos = np.zeros((n, 35))
    qvel = rng.randn(n, 34) * 2.0
    xipos = rng.randn(n, 14, 3) * 0.3
    xipos[:, :, 2] += rng.uniform(0.4, 2.3, size=(n, 1))  # COM z on both sides of [0.7, 2.0]
    xipos[:, 0, :] = 0.0
    idx = rng.randint(0, F, size=n)
    for e in range(n):
        qpos[e] = np.array(mocap.data_config[idx[e]]) + 0.2 * rng.randn(35)
    # pin a few exact threshold cases for is_done (z_com == 0.7 / 2.0 are NOT done: strict <, >)
    for e, z in zip(range(4), [0.7, 2.0, 0.69999, 2.00001]):
        xipos[e, 1:, :] = 0.0
        xipos[e, 1:, 2] = z

    obs = np.zeros((n, 56))
    done = np.zeros(n, dtype=np.uint8)
    zcom = np.zeros(n)
    rew_cfg = np.zeros(n)
    idx_after = np.zeros(n, dtype=np.int64)
    cfg_err = np.zeros(n)
    for e in range(n):
        env = E3.DPEnv.__new__(E3.DPEnv)
        env.mocap = mocap
        env.mocap_data_len = F
        env.model = FakeModel(body_mass)
        env.sim = FakeSim(FakeData(qpos[e], qvel[e], xipos[e]))
        env.idx_curr = int(idx[e])
        obs[e] = env._get_obs()
        done[e] = env.is_done()
        mass = np.expand_dims(env.model.body_mass, 1)
        zcom[e] = (np.sum(mass * env.sim.data.xipos, 0) / np.sum(mass))[2]
        cfg_err[e] = env.calc_config_errs(env.get_joint_configs(), mocap.data_config[env.idx_curr][7:])
        rew_cfg[e] = env.calc_config_reward()
        idx_after[e] = env.idx_curr

    # reference_state_init: Python global RNG stream (src/dp_env_v3.py:67-71)
    env = E3.DPEnv.__new__(E3.DPEnv)
    env.mocap_data_len = F
    random.seed(12345)
    rsi = []
    for _ in range(64):
        env.reference_state_init()
        rsi.append([env.idx_init, env.idx_curr, env.idx_tmp_count])

    # v2 reward (src/dp_env_v2.py:116-188): needs qpos[3:], idx_curr, idx_init, ctrl
    import dp_env_v2 as E2
    ctrl = rng.randn(n, 28) * 0.9
    idx_curr2 = rng.randint(0, 3 * F, size=n)
    idx_init2 = rng.randint(0, F, size=n)
    rew_v2 = np.zeros(n)
    rew_v2_total = np.zeros(n)
    root_err_v2 = np.zeros(n)
    vel_err_v2 = np.zeros(n)
    for e in range(n):
        env2 = E2.DPEnv.__new__(E2.DPEnv)
        env2.mocap = mocap
        env2.mocap_data_len = F
        env2.mocap_dt = mocap.dt
        env2.scale_err, env2.scale_pose = 1.0, 2.0
        env2.sim = FakeSim(FakeData(qpos[e], qvel[e], xipos[e], ctrl[e]))
        env2.idx_curr = int(idx_curr2[e])
        env2.idx_init = int(idx_init2[e])
        rew_v2[e] = env2.calc_reward()
        rew_v2_total[e] = rew_v2[e] - 0.1 * np.square(ctrl[e]).sum()  # src/dp_env_v2.py:181-183
        tgt = np.array(mocap.data[env2.idx_mocap, 1 + 2: 1 + 7])
        cur = np.array(qpos[e][2:7])
        cur[1:] /= np.linalg.norm(cur[1:])
        root_err_v2[e] = env2.calc_root_errs(cur, tgt)
        vel_err_v2[e] = env2.calc_vel_errs(mocap.data_vel[env2.idx_mocap][3:], qvel[e][3:])

    # MujocoInterface tables + v1 weighted pose error on quaternion poses (43 = 7*... see DOF_DEF)
    itf = MujocoInterface()
    npose = 64
    # pose vector in BODY_DEFS order incl. root quat: root(4) chest(4) neck(4) r_hip(4) r_knee(1) r_ankle(4)
    # r_shoulder(4) r_elbow(1) [r_wrist 0] l_hip(4) l_knee(1) l_ankle(4) l_shoulder(4) l_elbow(1) [l_wrist 0] = 40
    sizes = [4, 4, 4, 4, 1, 4, 4, 1, 0, 4, 1, 4, 4, 1, 0]
    L = sum(sizes)
    pa = np.zeros((npose, L))
    pb = np.zeros((npose, L))
    for arr in (pa, pb):
        o = 0
        for s in sizes:
            if s == 4:
                q = rng.randn(npose, 4)
                arr[:, o:o + 4] = q / np.linalg.norm(q, axis=1, keepdims=True)
            elif s == 1:
                arr[:, o] = rng.uniform(-2, 2, size=npose)
            o += s
    v1_pose_err = np.array([itf.calc_config_errs(a, b) for a, b in zip(pa, pb)])
    v1_err_vec = np.array([itf.calc_config_err_vec(a, b) for a, b in zip(pa, pb)])
    va = rng.randn(npose, 34)
    vb = rng.randn(npose, 34)
    v1_vel_err = np.array([itf.calc_vel_errs(a, b) for a, b in zip(va, vb)])
    ra = rng.randn(npose, 3)
    rb = rng.randn(npose, 3)
    v1_root_err = np.array([itf.calc_root_errs(a, b) for a, b in zip(ra, rb)])

    np.savez(os.path.join(OUT, "env_logic_golden.npz"),
             body_mass=np.array(body_mass, dtype=np.float64), qpos=qpos, qvel=qvel, xipos=xipos,
             idx=idx, obs=obs, done=done, zcom=zcom, cfg_err=cfg_err, rew_cfg=rew_cfg,
             idx_after=idx_after, n_frames=np.int64(F), rsi=np.array(rsi, dtype=np.int64),
             rsi_seed=np.int64(12345),
             ctrl=ctrl, idx_curr2=idx_curr2, idx_init2=idx_init2, rew_v2=rew_v2,
             rew_v2_total=rew_v2_total, root_err_v2=root_err_v2, vel_err_v2=vel_err_v2,
             kp=itf.kp.astype(np.float64), kd=itf.kd.astype(np.float64),
             pose_a=pa, pose_b=pb, v1_pose_err=v1_pose_err, v1_err_vec=v1_err_vec,
             vel_a=va, vel_b=vb, v1_vel_err=v1_vel_err, root_a=ra, root_b=rb, v1_root_err=v1_root_err)
    print("env logic: obs", obs.shape, "done frac %.2f" % done.mean(), "rew_cfg[:3]", rew_cfg[:3])


def main():
    if not os.path.isdir(REF):
        raise SystemExit("reference not present; fixtures can only be regenerated in the build container")
    os.makedirs(OUT, exist_ok=True)
    setup_imports()
    gen_mocap()
    gen_euler()
    gen_quat_ops()
    gen_env_logic()


if __name__ == "__main__":
    main()
