"""Host-side rotation helpers for the mocap pipeline (init-time only; numpy float64, batched).

These mirror, for arrays of shape [..., 4] / [..., 3], the handful of rotation operations the
reference performs one frame at a time:

* Hamilton product / conjugate / axis / wrapped angle of (w, x, y, z) quaternions — the
  pyquaternion operations used at src/mujoco/mocap_util.py:31-40,50-77 and
  src/mujoco/mocap_v2.py:64-76 (third-party, unpinned; semantics: `angle` normalises the quaternion
  and wraps 2*atan2(|v|, w) into (-pi, pi]; `axis` is v/|v| or zeros when |v| < 1e-17).
* `align_rotation` / `align_position`: the Y-up -> Z-up change of basis, q -> qL * q * qR with
  qL = Rx(+90deg), qR = Rx(-90deg) (src/mujoco/mocap_util.py:31-48).
* `euler_rxyz_from_quat_xyzw`: rotating-frame XYZ Euler angles of an [x, y, z, w] quaternion, i.e.
  `euler_from_quaternion(q, 'rxyz')` of src/transformations.py:1089-1097 (via quaternion_matrix
  :1174-1193 and euler_from_matrix :1031-1086 with _AXES2TUPLE['rxyz'] = (2, 1, 0, 1)).
"""
import numpy as np

_EPS = np.finfo(np.float64).eps * 4.0  # src/transformations.py:1515
_RSQRT2 = 2.0 * (0.5 / np.sqrt(2.0))   # the value the trace method yields for a 90deg rotation about x
_QL = np.array([_RSQRT2, _RSQRT2, 0.0, 0.0])   # Rx(+90deg)
_QR = np.array([_RSQRT2, -_RSQRT2, 0.0, 0.0])  # Rx(-90deg)


def quat_mul(a, b):
    """Hamilton product of (w,x,y,z) quaternions, broadcasting over leading dims."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    aw, ax, ay, az = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bw, bx, by, bz = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([aw * bw - ax * bx - ay * by - az * bz,
                     ax * bw + aw * bx - az * by + ay * bz,
                     ay * bw + az * bx + aw * by - ax * bz,
                     az * bw - ay * bx + ax * by + aw * bz], axis=-1)


def quat_conj(a):
    a = np.asarray(a, dtype=np.float64)
    return a * np.array([1.0, -1.0, -1.0, -1.0])


def _normalised(q):
    """pyquaternion's lazy normalisation: untouched when |1 - q.q| < 1e-14, else q/|q| (if |q|>0)."""
    q = np.asarray(q, dtype=np.float64)
    ss = np.sum(q * q, axis=-1, keepdims=True)
    n = np.sqrt(ss)
    need = (np.abs(1.0 - ss) >= 1e-14) & (n > 0)
    return np.where(need, q / np.where(n > 0, n, 1.0), q)


def quat_axis(q):
    q = _normalised(q)
    v = q[..., 1:4]
    n = np.linalg.norm(v, axis=-1, keepdims=True)
    return np.where(n < 1e-17, 0.0, v / np.where(n < 1e-17, 1.0, n))


def quat_angle(q):
    q = _normalised(q)
    n = np.linalg.norm(q[..., 1:4], axis=-1)
    theta = 2.0 * np.arctan2(n, q[..., 0])
    res = np.mod(theta + np.pi, 2.0 * np.pi) - np.pi
    return np.where(res == -np.pi, np.pi, res)


def align_rotation(q):
    """(w,x,y,z) in the Y-up mocap frame -> Z-up MuJoCo frame; equals (w, x, -z, y) up to rounding."""
    return quat_mul(quat_mul(_QL, q), _QR)


def align_position(p):
    p = np.asarray(p, dtype=np.float64)
    return np.stack([p[..., 0], -p[..., 2], p[..., 1]], axis=-1)


def rot_vel(q_a, q_b, dura):
    """angle(a* . b)/dura * axis(a* . b): `MocapDM.calc_rot_vel` (src/mujoco/mocap_v2.py:64-76)."""
    d = quat_mul(quat_conj(q_a), q_b)
    dura = np.asarray(dura, dtype=np.float64)
    return (quat_angle(d) / dura)[..., None] * quat_axis(d)


def angle_diff(q_a, q_b):
    """|relative rotation angle| between aligned quaternions, signed as pyquaternion's wrapped angle
    (src/mujoco/mocap_util.py:67-77: both operands go through align_rotation first)."""
    d = quat_mul(quat_conj(align_rotation(q_a)), align_rotation(q_b))
    return quat_angle(d)


def angular_vel_from_quat(q_a, q_b, dt):
    """src/mujoco/mocap_util.py:50-65 (aligns both operands, then angle/dt * axis)."""
    return rot_vel(align_rotation(q_a), align_rotation(q_b), dt)


def euler_rxyz_from_quat_xyzw(q):
    """Batched `euler_from_quaternion(q, axes='rxyz')` for q = [x, y, z, w] (src/transformations.py)."""
    q = np.array(q, dtype=np.float64, copy=True)
    nq = np.sum(q * q, axis=-1, keepdims=True)
    degenerate = (nq < _EPS)[..., 0]
    s = q * np.sqrt(2.0 / np.where(nq < _EPS, 1.0, nq))
    o = s[..., :, None] * s[..., None, :]
    M = np.empty(q.shape[:-1] + (3, 3))
    M[..., 0, 0] = 1.0 - o[..., 1, 1] - o[..., 2, 2]
    M[..., 0, 1] = o[..., 0, 1] - o[..., 2, 3]
    M[..., 0, 2] = o[..., 0, 2] + o[..., 1, 3]
    M[..., 1, 0] = o[..., 0, 1] + o[..., 2, 3]
    M[..., 1, 1] = 1.0 - o[..., 0, 0] - o[..., 2, 2]
    M[..., 1, 2] = o[..., 1, 2] - o[..., 0, 3]
    M[..., 2, 0] = o[..., 0, 2] - o[..., 1, 3]
    M[..., 2, 1] = o[..., 1, 2] + o[..., 0, 3]
    M[..., 2, 2] = 1.0 - o[..., 0, 0] - o[..., 1, 1]
    M[degenerate] = np.eye(3)
    # 'rxyz' -> firstaxis 2, parity 1, repetition 0, frame 1  => i=2, j=_NEXT_AXIS[3]=1, k=_NEXT_AXIS[2]=0
    i, j, k = 2, 1, 0
    cy = np.sqrt(M[..., i, i] ** 2 + M[..., j, i] ** 2)
    reg = cy > _EPS
    ax = np.where(reg, np.arctan2(M[..., k, j], M[..., k, k]), np.arctan2(-M[..., j, k], M[..., j, j]))
    ay = np.arctan2(-M[..., k, i], cy)
    az = np.where(reg, np.arctan2(M[..., j, i], M[..., i, i]), 0.0)
    ax, ay, az = -ax, -ay, -az      # parity
    ax, az = az, ax                  # rotating frame
    return np.stack([ax, ay, az], axis=-1)
