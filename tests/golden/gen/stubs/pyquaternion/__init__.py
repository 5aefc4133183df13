"""Stand-in for the six pyquaternion (~v0.9.5) operations the reference's mocap code uses.

TEST INFRASTRUCTURE ONLY (fixture generation inside the build container).  pyquaternion is a
third-party dependency of the reference (README.md:33, unpinned) that is not installed here, so
its published semantics are restated: Quaternion(w,x,y,z), Quaternion(matrix=R) (trace method on
R^T), .conjugate, Hamilton __mul__ (q_matrix . q), .elements, .axis, .angle (wrapped to (-pi,pi]).
Call sites in the reference: src/mujoco/mocap_util.py:31-40,50-77; src/mujoco/mocap_v2.py:64-76;
src/dp_env_v2.py:101-114.  Never imported by the product package.
"""
from math import atan2, pi, sqrt

import numpy as np


class Quaternion(object):
    def __init__(self, *args, **kwargs):
        if "matrix" in kwargs:
            self.q = self._from_matrix(np.asarray(kwargs["matrix"], dtype=np.float64))
        elif "scalar" in kwargs or "vector" in kwargs:
            v = kwargs.get("vector", (0.0, 0.0, 0.0))
            self.q = np.array([kwargs.get("scalar", 0.0), v[0], v[1], v[2]], dtype=np.float64)
        elif len(args) == 4:
            self.q = np.array([float(a) for a in args], dtype=np.float64)
        elif len(args) == 0:
            self.q = np.array([1.0, 0.0, 0.0, 0.0])
        else:
            raise ValueError("unsupported Quaternion constructor in stand-in")

    @staticmethod
    def _from_matrix(matrix):
        if matrix.shape != (3, 3):
            raise ValueError("stand-in supports 3x3 rotation matrices only")
        if not np.allclose(np.dot(matrix, matrix.conj().transpose()), np.eye(3), rtol=1e-5, atol=1e-8):
            raise ValueError("Matrix must be orthogonal")
        if not np.isclose(np.linalg.det(matrix), 1.0, rtol=1e-5, atol=1e-8):
            raise ValueError("Matrix must be special orthogonal")
        m = matrix.conj().transpose()
        if m[2, 2] < 0:
            if m[0, 0] > m[1, 1]:
                t = 1 + m[0, 0] - m[1, 1] - m[2, 2]
                q = [m[1, 2] - m[2, 1], t, m[0, 1] + m[1, 0], m[2, 0] + m[0, 2]]
            else:
                t = 1 - m[0, 0] + m[1, 1] - m[2, 2]
                q = [m[2, 0] - m[0, 2], m[0, 1] + m[1, 0], t, m[1, 2] + m[2, 1]]
        else:
            if m[0, 0] < -m[1, 1]:
                t = 1 - m[0, 0] - m[1, 1] + m[2, 2]
                q = [m[0, 1] - m[1, 0], m[2, 0] + m[0, 2], m[1, 2] + m[2, 1], t]
            else:
                t = 1 + m[0, 0] + m[1, 1] + m[2, 2]
                q = [t, m[1, 2] - m[2, 1], m[2, 0] - m[0, 2], m[0, 1] - m[1, 0]]
        q = np.array(q).astype("float64")
        q *= 0.5 / sqrt(t)
        return q

    # --- algebra -------------------------------------------------------------------------
    def _q_matrix(self):
        w, x, y, z = self.q
        return np.array([[w, -x, -y, -z], [x, w, -z, y], [y, z, w, -x], [z, -y, x, w]])

    def __mul__(self, other):
        if isinstance(other, Quaternion):
            r = Quaternion()
            r.q = np.dot(self._q_matrix(), other.q)
            return r
        raise TypeError("stand-in supports Quaternion*Quaternion only")

    @property
    def conjugate(self):
        return Quaternion(scalar=self.q[0], vector=-self.q[1:4])

    @property
    def elements(self):
        return self.q

    @property
    def scalar(self):
        return self.q[0]

    @property
    def vector(self):
        return self.q[1:4]

    def _sum_of_squares(self):
        return np.dot(self.q, self.q)

    @property
    def norm(self):
        return sqrt(self._sum_of_squares())

    def is_unit(self, tolerance=1e-14):
        return abs(1.0 - self._sum_of_squares()) < tolerance

    def _normalise(self):
        if not self.is_unit():
            n = self.norm
            if n > 0:
                self.q = self.q / n

    @staticmethod
    def _wrap_angle(theta):
        result = ((theta + pi) % (2 * pi)) - pi
        if result == -pi:
            result = pi
        return result

    @property
    def axis(self):
        tolerance = 1e-17
        self._normalise()
        norm = np.linalg.norm(self.vector)
        if norm < tolerance:
            return np.zeros(3)
        return self.vector / norm

    @property
    def angle(self):
        self._normalise()
        norm = np.linalg.norm(self.vector)
        return self._wrap_angle(2.0 * atan2(norm, self.scalar))
