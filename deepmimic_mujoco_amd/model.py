"""Host-side model compiler: spec (humanoid.py / mjcf.py) -> the flat constant tables the HIP kernels read.

This is the part of MuJoCo's model compiler + `mj_setConst` that the DeepMimic humanoid needs
(EXTERNAL behaviour, restated; the reference reaches it through `mujoco_env.MujocoEnv.__init__`,
src/dp_env_v3.py:59): geom frames from `fromto`, body mass/COM/inertia from geoms (`inertiafromgeom`,
dp_env_v3.xml:2), kinematic-tree address tables, the candidate contact-pair list in MuJoCo's contact
order, and the constraint-regularisation constants `dof_invweight0` / `body_invweight0` /
`stat.meaninertia` evaluated at `qpos0`.  Init-time only, numpy float64.

The mass matrix here is assembled as sum_b J_b^T diag(m, I_b) J_b from body Jacobians — deliberately a
different algorithm from the composite-rigid-body recursion in the HIP kernel and in the CPU oracle.
"""
import numpy as np

GEOM_TYPES = {"plane": 0, "sphere": 2, "capsule": 3, "box": 6}   # mjtGeom values
JNT_FREE, JNT_HINGE = 0, 3


def _quat_mat(q):
    w, x, y, z = q
    return np.array([[w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z]])


def _z_to_vec_quat(vec):
    """Rotation taking +z to `vec` (unit): axis = z x vec (or x when degenerate), angle = atan2(|z x vec|, vec_z)."""
    ax = np.cross([0.0, 0.0, 1.0], vec)
    s = np.linalg.norm(ax)
    ax = np.array([1.0, 0.0, 0.0]) if s < 1e-10 else ax / s
    ang = np.arctan2(s, vec[2])
    return np.concatenate([[np.cos(ang / 2)], ax * np.sin(ang / 2)])


def _axis_angle_mat(axis, angle):
    q = np.concatenate([[np.cos(angle / 2)], np.asarray(axis, dtype=np.float64) * np.sin(angle / 2)])
    return _quat_mat(q)


class CompiledModel(object):
    """Flat numpy tables; attribute names follow mjModel where a counterpart exists."""

    def __init__(self, spec):
        self.spec = spec
        o = spec["option"]
        if o.get("integrator", "RK4") != "RK4" or o.get("solver", "PGS") != "PGS" or o.get("cone", "pyramidal") != "pyramidal":
            raise ValueError("the accelerated path implements integrator=RK4, solver=PGS, cone=pyramidal (dp_env_v3.xml:9)")
        B, J, G, U = spec["bodies"], spec["joints"], spec["geoms"], spec["motors"]
        self.nbody, self.njnt, self.ngeom, self.nu = len(B), len(J), len(G), len(U)
        self.timestep = float(o["timestep"]); self.iterations = int(o["iterations"]); self.tolerance = float(o["tolerance"])
        self.gravity = np.array(o["gravity"], dtype=np.float64)
        self.solref = np.array(o["solref"], dtype=np.float64); self.solimp = np.array(o["solimp"], dtype=np.float64)
        self.body_parentid = np.array([b["parent"] for b in B], dtype=np.int32)
        self.body_pos = np.array([b["pos"] for b in B], dtype=np.float64)
        self.body_names = [b["name"] for b in B]
        # ---- joints / dofs -------------------------------------------------------------------------
        self.jnt_type = np.array([JNT_FREE if j["type"] == "free" else JNT_HINGE for j in J], dtype=np.int32)
        self.jnt_bodyid = np.array([j["body"] for j in J], dtype=np.int32)
        if np.any(np.diff(self.jnt_bodyid) < 0):
            raise ValueError("joints must be ordered by body")
        self.jnt_axis = np.array([j["axis"] for j in J], dtype=np.float64)
        self.jnt_range = np.array([j["range"] for j in J], dtype=np.float64)
        self.jnt_limited = np.array([1 if j["limited"] else 0 for j in J], dtype=np.int32)
        self.jnt_qposadr = np.zeros(self.njnt, dtype=np.int32); self.jnt_dofadr = np.zeros(self.njnt, dtype=np.int32)
        nq = nv = 0
        dof_body, dof_jnt, arm, damp = [], [], [], []
        for i, j in enumerate(J):
            self.jnt_qposadr[i], self.jnt_dofadr[i] = nq, nv
            dq, dv = (7, 6) if j["type"] == "free" else (1, 1)
            dof_body += [j["body"]] * dv; dof_jnt += [i] * dv
            arm += [j["armature"]] * dv; damp += [j["damping"]] * dv
            nq += dq; nv += dv
        self.nq, self.nv = nq, nv
        self.dof_bodyid = np.array(dof_body, dtype=np.int32); self.dof_jntid = np.array(dof_jnt, dtype=np.int32)
        self.dof_armature = np.array(arm, dtype=np.float64); self.dof_damping = np.array(damp, dtype=np.float64)
        self.body_dofadr = np.full(self.nbody, -1, dtype=np.int32); self.body_dofnum = np.zeros(self.nbody, dtype=np.int32)
        for d, b in enumerate(dof_body):
            if self.body_dofnum[b] == 0:
                self.body_dofadr[b] = d
            self.body_dofnum[b] += 1
        self.dof_parentid = np.full(nv, -1, dtype=np.int32)
        for d in range(nv):
            b = dof_body[d]
            if d > self.body_dofadr[b]:
                self.dof_parentid[d] = d - 1
                continue
            p = self.body_parentid[b]
            while p > 0 and self.body_dofnum[p] == 0:
                p = self.body_parentid[p]
            if p > 0:
                self.dof_parentid[d] = self.body_dofadr[p] + self.body_dofnum[p] - 1
        self.qpos0 = np.zeros(nq)
        for i, j in enumerate(J):
            if j["type"] == "free":
                a = self.jnt_qposadr[i]
                self.qpos0[a:a + 3] = self.body_pos[j["body"]]; self.qpos0[a + 3] = 1.0
        # ---- geoms ---------------------------------------------------------------------------------
        self.geom_type = np.array([GEOM_TYPES[g["type"]] for g in G], dtype=np.int32)
        self.geom_bodyid = np.array([g["body"] for g in G], dtype=np.int32)
        self.geom_condim = np.array([g["condim"] for g in G], dtype=np.int32)
        self.geom_contype = np.array([g["contype"] for g in G], dtype=np.int32)
        self.geom_conaffinity = np.array([g["conaffinity"] for g in G], dtype=np.int32)
        self.geom_friction = np.array([g["friction"] for g in G], dtype=np.float64)
        self.geom_margin = np.array([g["margin"] for g in G], dtype=np.float64)
        self.geom_mass = np.array([g["mass"] for g in G], dtype=np.float64)
        self.geom_size = np.zeros((self.ngeom, 3)); self.geom_pos = np.zeros((self.ngeom, 3))
        self.geom_quat = np.tile([1.0, 0, 0, 0], (self.ngeom, 1)); self.geom_mat = np.zeros((self.ngeom, 3, 3))
        for i, g in enumerate(G):
            sz = list(g["size"]) + [0.0] * (3 - len(g["size"]))
            if g["fromto"] is not None:
                a, b = np.array(g["fromto"][:3]), np.array(g["fromto"][3:])
                ln = np.linalg.norm(b - a)
                self.geom_pos[i] = 0.5 * (a + b); sz[1] = 0.5 * ln
                self.geom_quat[i] = _z_to_vec_quat((b - a) / ln)
            else:
                self.geom_pos[i] = g["pos"]
            self.geom_size[i] = sz[:3]
            self.geom_mat[i] = _quat_mat(self.geom_quat[i])
        # ---- body inertial properties from geoms ---------------------------------------------------
        self.body_mass = np.zeros(self.nbody); self.body_ipos = np.zeros((self.nbody, 3))
        self.body_inertia = np.zeros((self.nbody, 3, 3))   # about the COM, body-frame axes (full symmetric 3x3)
        for b in range(self.nbody):
            gs = [i for i in range(self.ngeom) if self.geom_bodyid[i] == b and self.geom_mass[i] > 0]
            if not gs:
                continue
            mtot = self.geom_mass[gs].sum()
            com = (self.geom_mass[gs, None] * self.geom_pos[gs]).sum(0) / mtot
            I = np.zeros((3, 3))
            for i in gs:
                I += self._geom_inertia(i, com)
            self.body_mass[b], self.body_ipos[b], self.body_inertia[b] = mtot, com, I
        self.total_mass = self.body_mass.sum()
        # ---- actuators -----------------------------------------------------------------------------
        self.actuator_jntid = np.array([u["joint"] for u in U], dtype=np.int32)
        self.actuator_dofid = self.jnt_dofadr[self.actuator_jntid] if U else np.zeros(0, dtype=np.int32)
        self.actuator_gear = np.array([u["gear"] for u in U], dtype=np.float64)
        self.actuator_ctrlrange = np.array([u["ctrlrange"] for u in U], dtype=np.float64).reshape(-1, 2)
        # ---- candidate contact pairs, in MuJoCo's contact-list order ------------------------------
        excl = set(tuple(sorted(e)) for e in spec["excludes"])
        pairs = []
        for b1 in range(self.nbody):
            for b2 in range(b1 + 1, self.nbody):
                if b1 != 0 and (self.body_parentid[b2] == b1 or self.body_parentid[b1] == b2):
                    continue
                if (b1, b2) in excl:
                    continue
                for g1 in np.nonzero(self.geom_bodyid == b1)[0]:
                    for g2 in np.nonzero(self.geom_bodyid == b2)[0]:
                        if not ((self.geom_contype[g1] & self.geom_conaffinity[g2]) or (self.geom_contype[g2] & self.geom_conaffinity[g1])):
                            continue
                        if self.geom_type[g1] == 0 and self.geom_type[g2] == 0:
                            continue
                        a, c = (g1, g2) if self.geom_type[g1] <= self.geom_type[g2] else (g2, g1)
                        pairs.append((int(a), int(c)))
        self.pair_geom = np.array(pairs, dtype=np.int32).reshape(-1, 2)
        self.npair = len(pairs)
        self._set_const()

    def _geom_inertia(self, i, com):
        m, t, s = self.geom_mass[i], self.geom_type[i], self.geom_size[i]
        if t == 2:
            d = np.full(3, 0.4 * m * s[0] ** 2)
        elif t == 6:
            d = m / 3.0 * np.array([s[1] ** 2 + s[2] ** 2, s[0] ** 2 + s[2] ** 2, s[0] ** 2 + s[1] ** 2])
        elif t == 3:
            r, h = s[0], s[1]
            vs, vc = 4.0 / 3.0 * r, 2.0 * h
            ms = m * vs / (vs + vc); mc = m - ms
            izz = mc * r * r / 2 + 0.4 * ms * r * r
            ixx = mc * (3 * r * r + 4 * h * h) / 12 + ms * (0.4 * r * r + h * h + 0.75 * r * h)
            d = np.array([ixx, ixx, izz])
        else:
            d = np.zeros(3)
        R = self.geom_mat[i]
        rr = self.geom_pos[i] - com
        return R @ np.diag(d) @ R.T + m * (rr @ rr * np.eye(3) - np.outer(rr, rr))

    # ---- kinematics / Jacobians at an arbitrary qpos (host side, used for set-const and by tests) ------
    def kinematics(self, qpos):
        xpos = np.zeros((self.nbody, 3)); xmat = np.tile(np.eye(3), (self.nbody, 1, 1))
        axes = np.zeros((self.nv, 3)); anchors = np.zeros((self.nv, 3)); is_rot = np.zeros(self.nv, dtype=bool)
        for b in range(1, self.nbody):
            p = self.body_parentid[b]
            js = np.nonzero(self.jnt_bodyid == b)[0]
            if len(js) == 1 and self.jnt_type[js[0]] == JNT_FREE:
                qa, da = self.jnt_qposadr[js[0]], self.jnt_dofadr[js[0]]
                xpos[b] = qpos[qa:qa + 3]
                q = qpos[qa + 3:qa + 7] / np.linalg.norm(qpos[qa + 3:qa + 7])
                xmat[b] = _quat_mat(q)
                for k in range(3):
                    axes[da + k] = np.eye(3)[k]
                    axes[da + 3 + k] = xmat[b][:, k]; anchors[da + 3 + k] = xpos[b]; is_rot[da + 3 + k] = True
                continue
            xpos[b] = xpos[p] + xmat[p] @ self.body_pos[b]
            R = xmat[p].copy()
            for j in js:
                qa, da = self.jnt_qposadr[j], self.jnt_dofadr[j]
                axes[da] = R @ self.jnt_axis[j]; anchors[da] = xpos[b]; is_rot[da] = True
                R = R @ _axis_angle_mat(self.jnt_axis[j], qpos[qa] - self.qpos0[qa])
            xmat[b] = R
        xipos = xpos + np.einsum("bij,bj->bi", xmat, self.body_ipos)
        return xpos, xmat, xipos, axes, anchors, is_rot

    def body_jacobian(self, b, point, axes, anchors, is_rot):
        jp = np.zeros((3, self.nv)); jr = np.zeros((3, self.nv))
        while b > 0 and self.body_dofnum[b] == 0:
            b = self.body_parentid[b]
        if b <= 0:
            return jp, jr
        d = self.body_dofadr[b] + self.body_dofnum[b] - 1
        while d >= 0:
            if is_rot[d]:
                jr[:, d] = axes[d]; jp[:, d] = np.cross(axes[d], point - anchors[d])
            else:
                jp[:, d] = axes[d]
            d = self.dof_parentid[d]
        return jp, jr

    def mass_matrix(self, qpos):
        xpos, xmat, xipos, axes, anchors, is_rot = self.kinematics(qpos)
        M = np.diag(self.dof_armature.copy())
        for b in range(1, self.nbody):
            if self.body_mass[b] <= 0:
                continue
            jp, jr = self.body_jacobian(b, xipos[b], axes, anchors, is_rot)
            Iw = xmat[b] @ self.body_inertia[b] @ xmat[b].T
            M += self.body_mass[b] * jp.T @ jp + jr.T @ Iw @ jr
        return M

    def _set_const(self):
        xpos, xmat, xipos, axes, anchors, is_rot = self.kinematics(self.qpos0)
        M = self.mass_matrix(self.qpos0)
        Minv = np.linalg.inv(M)
        self.meaninertia = float(np.mean(np.diag(M))) if self.nv else 1.0
        self.dof_invweight0 = np.diag(Minv).copy()
        for j in range(self.njnt):
            if self.jnt_type[j] == JNT_FREE:
                a = self.jnt_dofadr[j]
                self.dof_invweight0[a:a + 3] = np.diag(Minv)[a:a + 3].mean()
                self.dof_invweight0[a + 3:a + 6] = np.diag(Minv)[a + 3:a + 6].mean()
        self.body_invweight0 = np.zeros((self.nbody, 2))
        for b in range(1, self.nbody):
            jp, jr = self.body_jacobian(b, xipos[b], axes, anchors, is_rot)
            self.body_invweight0[b, 0] = np.trace(jp @ Minv @ jp.T) / 3.0
            self.body_invweight0[b, 1] = np.trace(jr @ Minv @ jr.T) / 3.0
        self.init_com_z = float((self.body_mass * xipos[:, 2]).sum() / self.total_mass) if self.total_mass > 0 else 0.0
