"""The full 5-term DeepMimic imitation reward: host-side feature tables.

Specification: `cSceneImitate::CalcRewardImitate` as quoted in the reference's porting notes (`code.md:1017-1143`) — the
reference itself never implemented it (`dp_env_v3.py:117-128` returns 1.0).  The terms

    r = 0.5 exp(-2 pose_err) + 0.05 exp(-0.1 vel_err) + 0.15 exp(-40 end_eff_err) + 0.2 exp(-5 root_err) + 0.1 exp(-10 com_err)

compare a *feature vector* of the simulated character with the same features of the mocap frame it should be at.  This
module defines the feature vector once, in numpy, for any (qpos, qvel) of the humanoid model; `build_table` evaluates it on
every mocap frame (the device keeps the [F, 112] table next to the mocap arrays), and the kernel epilogue / the oracle
evaluate the same features on the simulated state.  Layout of a feature row (FEAT = 112 doubles):

    0:3    root position (world; ground height 0)          3:7   root quaternion (w, x, y, z)
    7:10   root linear velocity (world)                    10:13 root angular velocity (world)
    13:61  12 joints x 4: rotation of the child frame relative to its parent as a quaternion (3-hinge joints, hinge order
           as in the model), or (angle, 0, 0, 0) for the 1-hinge joints (knees, elbows)
    61:97  12 joints x 3: angular velocity of the child relative to the parent, parent frame; (rate, 0, 0) for 1-hinge joints
    97:109 4 end effectors x 3 (right ankle, left ankle, right wrist, left wrist): position relative to the root, height
           above ground in z, rotated into the root's heading frame                         (code.md:1085-1100)
    109:112 centre-of-mass velocity (world)

Joint order = model body order (chest, neck, right_shoulder, right_elbow, left_shoulder, left_elbow, right_hip, right_knee,
right_ankle, left_hip, left_knee, left_ankle).  Items restated from upstream DeepMimic (KinTree / MathUtil: squared
quaternion-difference angle per joint, L1-normalised joint weights) are not in the reference's files; they are marked
`[upstream]`.  Coordinates are the MuJoCo model's (z up, x forward).
"""
import numpy as np

FEAT = 112
O_RPOS, O_RQUAT, O_RLIN, O_RANG, O_JQ, O_JW, O_EE, O_COMV = 0, 3, 7, 10, 13, 61, 97, 109
NJ, NEE = 12, 4

# DiffWeight per joint: src/data/characters/humanoid3d.txt (Skeleton.Joints[*].DiffWeight), keyed by the model's body names
DIFF_WEIGHT = {"root": 1.0, "chest": 0.5, "neck": 0.3, "right_hip": 0.5, "right_knee": 0.3, "right_ankle": 0.2,
               "right_shoulder": 0.3, "right_elbow": 0.2, "left_hip": 0.5, "left_knee": 0.3, "left_ankle": 0.2,
               "left_shoulder": 0.3, "left_elbow": 0.2}
# IsEndEffector joints of the same file: ankles (5, 11) and wrists (8, 14; fixed joints at AttachY = -0.258947 of the elbows)
END_EFFECTORS = (("right_ankle", (0.0, 0.0, 0.0)), ("left_ankle", (0.0, 0.0, 0.0)),
                 ("right_elbow", (0.0, 0.0, -0.258947)), ("left_elbow", (0.0, 0.0, -0.258947)))
# term weights and scales: code.md:1019-1037
TERM_W = np.array([0.5, 0.05, 0.15, 0.2, 0.1])
TERM_SCALE = np.array([2.0, 0.1, 40.0, 5.0, 10.0])


def quat_mul(a, b):
    return np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
                     a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                     a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
                     a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]])


def quat_rot(q, v):
    w, x, y, z = q
    Rm = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                   [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                   [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    return Rm @ v


def axis_quat(axis, angle):
    h = 0.5 * angle
    return np.concatenate([[np.cos(h)], np.sin(h) * np.asarray(axis, dtype=np.float64)])


def quat_diff_theta(q0, q1):
    """[upstream] cMathUtil::QuatDiffTheta: rotation angle of q1 * conj(q0), normalised to [-pi, pi]."""
    d = quat_mul(q1, np.array([q0[0], -q0[1], -q0[2], -q0[3]]))
    w = min(1.0, max(-1.0, d[0]))
    if np.sqrt(max(0.0, 1.0 - w * w)) <= 1e-6:
        return 0.0
    th = 2.0 * np.arccos(w)
    return th - 2.0 * np.pi if th > np.pi else th


class ImitationSpec:
    """Static description of the features for one compiled model: joint groups, weights, end effectors."""

    def __init__(self, cm, normalize_weights=True):
        self.cm = cm
        names = list(cm.body_names)
        self.bodies = [b for b in range(2, cm.nbody)]                 # every body below the root carries one joint group
        assert len(self.bodies) == NJ
        w = np.array([DIFF_WEIGHT[names[b]] for b in self.bodies])
        w_root = DIFF_WEIGHT["root"]
        if normalize_weights:                                          # [upstream] CalcJointWeights: weights / L1 sum (wrists weigh 0)
            tot = w.sum() + w_root
            w, w_root = w / tot, w_root / tot
        self.w_joint, self.w_root = w, w_root
        self.ee_body = np.array([names.index(n) for n, _ in END_EFFECTORS], dtype=np.int32)
        self.ee_off = np.array([o for _, o in END_EFFECTORS], dtype=np.float64)

    def features(self, qpos, qvel):
        cm = self.cm
        f = np.zeros(FEAT)
        q = np.asarray(qpos, dtype=np.float64); v = np.asarray(qvel, dtype=np.float64)
        rq = q[3:7] / np.linalg.norm(q[3:7])
        f[O_RPOS:O_RPOS + 3] = q[0:3]
        f[O_RQUAT:O_RQUAT + 4] = rq
        f[O_RLIN:O_RLIN + 3] = v[0:3]
        f[O_RANG:O_RANG + 3] = quat_rot(rq, v[3:6])                    # free-joint angular velocity is body-local in MuJoCo
        for g, b in enumerate(self.bodies):
            js = np.nonzero(cm.jnt_bodyid == b)[0]
            if len(js) == 1:
                f[O_JQ + 4 * g] = q[cm.jnt_qposadr[js[0]]]
                f[O_JW + 3 * g] = v[cm.jnt_dofadr[js[0]]]
            else:
                ql = np.array([1.0, 0, 0, 0]); wl = np.zeros(3)
                for j in js:                                            # child = R1 R2 R3; w = sum_k R1..R(k-1) a_k rate_k
                    ax = cm.jnt_axis[j]
                    wl = wl + quat_rot(ql, ax) * v[cm.jnt_dofadr[j]]
                    ql = quat_mul(ql, axis_quat(ax, q[cm.jnt_qposadr[j]]))
                f[O_JQ + 4 * g:O_JQ + 4 * g + 4] = ql
                f[O_JW + 3 * g:O_JW + 3 * g + 3] = wl
        xpos, xmat, xipos, axes, anchors, is_rot = cm.kinematics(q)
        fwd = quat_rot(rq, np.array([1.0, 0, 0]))
        hd = np.arctan2(fwd[1], fwd[0])                                 # heading about the vertical
        c, s_ = np.cos(hd), np.sin(hd)
        Rinv = np.array([[c, s_, 0], [-s_, c, 0], [0, 0, 1]])
        for e in range(NEE):
            p = xpos[self.ee_body[e]] + xmat[self.ee_body[e]] @ self.ee_off[e]
            rel = p - q[0:3]
            rel[2] = p[2]                                               # height above the ground plane (z = 0)
            f[O_EE + 3 * e:O_EE + 3 * e + 3] = Rinv @ rel
        mom = np.zeros(3)
        for b in range(1, cm.nbody):
            jp, _jr = cm.body_jacobian(b, xipos[b], axes, anchors, is_rot)
            mom += cm.body_mass[b] * (jp @ v)
        f[O_COMV:O_COMV + 3] = mom / cm.body_mass[1:].sum()
        return f

    def reward_terms(self, f0, f1, root_shift=(0.0, 0.0)):
        """(pose_err, vel_err, end_eff_err, root_err, com_err) of simulated features f0 against reference features f1 whose root
        has advanced by `root_shift` (x, y) through completed motion cycles.   code.md:1067-1127"""
        th_root = quat_diff_theta(f0[O_RQUAT:O_RQUAT + 4], f1[O_RQUAT:O_RQUAT + 4])
        dw_root = f1[O_RANG:O_RANG + 3] - f0[O_RANG:O_RANG + 3]
        pose_err = self.w_root * th_root ** 2
        vel_err = self.w_root * dw_root.dot(dw_root)
        for g, b in enumerate(self.bodies):
            if self.cm.body_dofnum[b] == 1:
                pe = (f1[O_JQ + 4 * g] - f0[O_JQ + 4 * g]) ** 2
            else:
                pe = quat_diff_theta(f0[O_JQ + 4 * g:O_JQ + 4 * g + 4], f1[O_JQ + 4 * g:O_JQ + 4 * g + 4]) ** 2
            dv = f1[O_JW + 3 * g:O_JW + 3 * g + 3] - f0[O_JW + 3 * g:O_JW + 3 * g + 3]
            pose_err += self.w_joint[g] * pe
            vel_err += self.w_joint[g] * dv.dot(dv)
        de = f1[O_EE:O_EE + 12] - f0[O_EE:O_EE + 12]
        end_eff_err = de.dot(de) / NEE
        p1 = f1[O_RPOS:O_RPOS + 3].copy(); p1[0] += root_shift[0]; p1[1] += root_shift[1]
        dp = f0[O_RPOS:O_RPOS + 3] - p1
        dv = f1[O_RLIN:O_RLIN + 3] - f0[O_RLIN:O_RLIN + 3]
        root_err = dp.dot(dp) + 0.1 * th_root ** 2 + 0.01 * dv.dot(dv) + 0.001 * dw_root.dot(dw_root)
        dc = f1[O_COMV:O_COMV + 3] - f0[O_COMV:O_COMV + 3]
        com_err = 0.1 * dc.dot(dc)
        return np.array([pose_err, vel_err, end_eff_err, root_err, com_err])

    def reward(self, f0, f1, root_shift=(0.0, 0.0)):
        w = TERM_W / TERM_W.sum()
        return float((w * np.exp(-TERM_SCALE * self.reward_terms(f0, f1, root_shift))).sum())

    def v1_reward_terms(self, f0, f1, f1v):
        """(err_pose, err_vel, err_root) of `dp_env_v1.calc_reward` (src/dp_env_v1.py:82-141) on this model's features:
        err_pose = `MujocoInterface.calc_config_errs` (src/mujoco/mujoco_interface.py:169-190): sum over the root and the 12 joints
        of JOINT_WEIGHT (mocap_util.py:26-29, un-normalised; = w / w_root here) x |rotation angle of q_sim^* q_ref| (|angle
        difference| for the 1-hinge joints); err_vel = `calc_vel_errs` (:205-210): L1 distance between the simulated angular rates
        (root, then joints) and the clip's rates from frame k to k + 1 (`f1v` = the feature row that carries them: row k + 1, or the
        last row); err_root = `calc_root_errs` (:192-199): L1 distance of the root positions (no cycle shift: v1 wraps the frame
        index only).  The v1 environment's own model has ball joints and a 43-number quaternion pose; this is the same reward on the
        hinge-triple model of dp_env_v3 (child-in-parent rotations composed from the triples)."""
        wr = self.w_root
        th_root = quat_diff_theta(f0[O_RQUAT:O_RQUAT + 4], f1[O_RQUAT:O_RQUAT + 4])
        err_pose = abs(th_root)
        err_vel = np.abs(f1v[O_RANG:O_RANG + 3] - f0[O_RANG:O_RANG + 3]).sum()
        for g, b in enumerate(self.bodies):
            if self.cm.body_dofnum[b] == 1:
                pe = abs(f1[O_JQ + 4 * g] - f0[O_JQ + 4 * g])
            else:
                pe = abs(quat_diff_theta(f0[O_JQ + 4 * g:O_JQ + 4 * g + 4], f1[O_JQ + 4 * g:O_JQ + 4 * g + 4]))
            err_pose += self.w_joint[g] / wr * pe
            err_vel += np.abs(f1v[O_JW + 3 * g:O_JW + 3 * g + 3] - f0[O_JW + 3 * g:O_JW + 3 * g + 3]).sum()
        err_root = np.abs(f0[O_RPOS:O_RPOS + 3] - f1[O_RPOS:O_RPOS + 3]).sum()
        return np.array([err_pose, err_vel, err_root])

    def v1_reward(self, f0, f1, f1v, ctrl=None):
        """0.5 e^(-2 err_pose) + 0.05 e^(-0.1 err_vel) + 0.2 e^(-5 err_root) [- 0.1 sum ctrl^2]   (src/dp_env_v1.py:42-53,130-139,147-150)"""
        e = self.v1_reward_terms(f0, f1, f1v)
        r = 0.5 * np.exp(-2.0 * e[0]) + 0.05 * np.exp(-0.1 * e[1]) + 0.2 * np.exp(-5.0 * e[2])
        return float(r - (0.1 * np.square(ctrl).sum() if ctrl is not None else 0.0))

    def reference_qvel(self, data_config, dura, loop="none"):
        """[F, 34] generalized velocity of the clip with the signs a tracking character must reproduce.

        `MocapDM.data_vel` keeps the reference's `calc_rot_vel(current, previous)` quirk (src/mujoco/mocap_v2.py:64-76,113,135):
        root and 3-hinge angular rates are the NEGATED backward difference, written as an axis-angle rate into Euler-hinge slots.
        That table is what RSI must copy into qvel (parity with `reset_model`), but as a reward TARGET it would penalise correct
        tracking.  Here frame k gets the backward difference (k-1 -> k) over `dura[k]` in the model's own coordinates:
        root linear as in the reference; root angular = rotation vector of conj(q_{k-1}) q_k (body-local, MuJoCo's free-joint
        convention); 1-hinge joints the angle difference; 3-hinge joints the hinge RATES that produce the child-in-parent
        angular velocity rotvec(ql_k conj(ql_{k-1})) / dura (3x3 solve against the hinge axes in the parent frame).
        Frame 0 takes the last frame's value on a looping clip (same pose one cycle later), frame 1's otherwise."""
        cm = self.cm
        cfg = np.asarray(data_config, dtype=np.float64)
        F = cfg.shape[0]
        v = np.zeros((F, cm.nv))
        if F < 2:
            return v

        def rotvec(dq):
            dq = dq / np.linalg.norm(dq)
            if dq[0] < 0:
                dq = -dq
            s = np.linalg.norm(dq[1:])
            return np.zeros(3) if s < 1e-300 else dq[1:] * (2.0 * np.arctan2(s, dq[0]) / s)

        def conj(q):
            return np.array([q[0], -q[1], -q[2], -q[3]])

        def local_chain(q, js):      # child-in-parent quaternion and the hinge axes in the parent frame
            ql = np.array([1.0, 0, 0, 0]); cols = []
            for j in js:
                ax = cm.jnt_axis[j]
                cols.append(quat_rot(ql, ax))
                ql = quat_mul(ql, axis_quat(ax, q[cm.jnt_qposadr[j]]))
            return ql, np.stack(cols, 1)

        groups = [np.nonzero(cm.jnt_bodyid == b)[0] for b in self.bodies]
        for k in range(1, F):
            dt = float(dura[k])
            v[k, 0:3] = (cfg[k, 0:3] - cfg[k - 1, 0:3]) / dt
            v[k, 3:6] = rotvec(quat_mul(conj(cfg[k - 1, 3:7]), cfg[k, 3:7])) / dt
            for js in groups:
                if len(js) == 1:
                    a = cm.jnt_qposadr[js[0]]
                    v[k, cm.jnt_dofadr[js[0]]] = (cfg[k, a] - cfg[k - 1, a]) / dt
                else:
                    q1, A1 = local_chain(cfg[k], js)
                    q0, _ = local_chain(cfg[k - 1], js)
                    w = rotvec(quat_mul(q1, conj(q0))) / dt
                    v[k, [cm.jnt_dofadr[j] for j in js]] = np.linalg.lstsq(A1, w, rcond=1e-9)[0]
        v[0] = v[F - 1] if str(loop) == "wrap" else v[1]
        return v

    def build_table(self, data_config, qvel):
        """[F, FEAT] reference features, one row per mocap frame.  `qvel` = `reference_qvel(...)` (correctly signed rates);
        passing `MocapDM.data_vel` reproduces the reference's sign quirk in the velocity features (round-1 behaviour)."""
        return np.stack([self.features(data_config[k], qvel[k]) for k in range(len(data_config))])

    def table_for(self, mocap):
        """(table [F, FEAT], params [32]) for a loaded `MocapDM`: what `Batch(imitation=...)` takes."""
        qv = self.reference_qvel(mocap.data_config, np.asarray(mocap.data)[:, 0], mocap.loop)
        return self.build_table(mocap.data_config, qv), self.params(mocap.data_config, mocap.loop)

    def params(self, data_config, loop):
        """The 32 doubles `dm_mocap_set_imitation` takes: joint weights [12], root weight, cycle shift (x, y), loop flag,
        end-effector body ids [4], end-effector offsets [4 x 3]."""
        p = np.zeros(32)
        p[0:12] = self.w_joint; p[12] = self.w_root
        p[13:15] = data_config[-1][0:2] - data_config[0][0:2]          # root advance of one motion cycle
        p[15] = 1.0 if str(loop) == "wrap" else 0.0
        p[16:20] = self.ee_body
        p[20:32] = self.ee_off.reshape(-1)
        return p
