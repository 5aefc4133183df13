"""Python front-end of the wave testbench (TEST INFRASTRUCTURE): runs the HIP kernel source on CPU fibres.
Presents the same small interface as deepmimic_mujoco_amd.batch.Batch so parity tests can drive either."""
import ctypes as C
import os
import subprocess

import numpy as np

from deepmimic_mujoco_amd import _abi as A

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
        L = C.CDLL(os.path.join(_HERE, "libdmemu.so"))
        L.emu_create.restype = C.c_void_p
        L.emu_create.argtypes = [C.POINTER(A.ModelDesc), A._dp, A._dp, C.c_int, C.c_int, C.c_uint, C.c_double]
        L.emu_destroy.argtypes = [C.c_void_p]
        L.emu_invalidate_kin.argtypes = [C.c_void_p]
        L.emu_set_option.argtypes = [C.c_void_p, C.c_int, C.c_longlong]
        L.emu_set_imitation.argtypes = [C.c_void_p, A._dp, A._dp]
        L.emu_field.restype = C.c_void_p
        L.emu_field.argtypes = [C.c_void_p, C.c_int]
        L.emu_step.argtypes = [C.c_void_p, A._dp, A._dp, A._dp, C.POINTER(C.c_uint8), C.c_int]
        L.emu_set_state.argtypes = [C.c_void_p, A._dp, A._dp, A._ip, C.POINTER(C.c_uint8)]
        L.emu_reset.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_uint8)]
        L.emu_redo_total.argtypes = [C.c_void_p]; L.emu_redo_total.restype = C.c_long
        L.emu_rollout.argtypes = [C.c_void_p, A._dp, A._dp, A._dp, C.POINTER(C.c_uint8), C.c_int, C.c_int]
        L.emu_debug_forward.argtypes = [C.c_void_p, C.c_int, A._dp]
        L.emu_dispatch.argtypes = [C.c_int, C.c_int, C.c_int, A._ip, A._ip, A._ip, C.c_int, A._ip, C.c_int, A._ip, A._ip]
        _LIB = L
    return _LIB


class EmuBatch(object):
    def __init__(self, cm, data_config, data_vel, n_envs, flags=0, imitation=None, mocap_dt=0.033332):
        self.n = n_envs
        md, self._keep = A.make_model_desc(cm)
        cfg = np.ascontiguousarray(data_config, dtype=np.float64); vel = np.ascontiguousarray(data_vel, dtype=np.float64)
        self.h = lib().emu_create(C.byref(md), cfg.ctypes.data_as(A._dp), vel.ctypes.data_as(A._dp), cfg.shape[0], n_envs, flags, float(mocap_dt))
        if not self.h:
            raise RuntimeError("emu_create failed")
        if imitation is not None:
            tab = np.ascontiguousarray(imitation[0], dtype=np.float64); par = np.ascontiguousarray(imitation[1], dtype=np.float64)
            lib().emu_set_imitation(self.h, tab.ctypes.data_as(A._dp), par.ctypes.data_as(A._dp))

    def __del__(self):
        if getattr(self, "h", None):
            lib().emu_destroy(self.h); self.h = None

    def set_option(self, opt, value):
        lib().emu_set_option(self.h, opt, int(value))

    def _view(self, field):
        dt, shp = A.FIELD_SPEC[field]
        p = lib().emu_field(self.h, field)
        n = int(np.prod((self.n,) + shp))
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double if dt == np.float64 else C.c_int32)), shape=(n,)).reshape((self.n,) + shp)

    def get(self, field):
        return self._view(field).copy()

    def set(self, field, value):
        self._view(field)[...] = np.asarray(value).reshape(self._view(field).shape)
        if field == A.F_QPOS:
            lib().emu_invalidate_kin(self.h)          # as dm_batch_set does: parked kinematics belong to the old positions

    def step(self, action, n_substeps=1, out=None):
        a = np.ascontiguousarray(action, dtype=np.float64).reshape(self.n, A.NU)
        if out is None:
            obs = np.zeros((self.n, A.NOBS)); rew = np.zeros(self.n); done = np.zeros(self.n, dtype=np.uint8)
        else:
            obs, rew, done = out
            assert obs.dtype == np.float64 and rew.dtype == np.float64 and done.dtype == np.uint8
            assert obs.flags.c_contiguous and obs.shape == (self.n, A.NOBS)
        lib().emu_step(self.h, a.ctypes.data_as(A._dp), obs.ctypes.data_as(A._dp), rew.ctypes.data_as(A._dp),
                       done.ctypes.data_as(C.POINTER(C.c_uint8)), n_substeps)
        return obs, rew, done

    def rollout(self, actions, n_substeps=1):
        """dm_batch_rollout on the packed path, open loop: actions [T, n, 28] -> (obs [T, n, 56], rew [T, n], done [T, n])"""
        a = np.ascontiguousarray(actions, dtype=np.float64)
        T = a.shape[0]
        assert a.shape == (T, self.n, A.NU)
        obs = np.zeros((T, self.n, A.NOBS)); rew = np.zeros((T, self.n)); done = np.zeros((T, self.n), dtype=np.uint8)
        lib().emu_rollout(self.h, a.ctypes.data_as(A._dp), obs.ctypes.data_as(A._dp), rew.ctypes.data_as(A._dp),
                          done.ctypes.data_as(C.POINTER(C.c_uint8)), n_substeps, T)
        return obs, rew, done

    def set_state(self, qpos, qvel, frame_idx=None, mask=None):
        q = np.ascontiguousarray(qpos, dtype=np.float64).reshape(self.n, A.NQ)
        v = np.ascontiguousarray(qvel, dtype=np.float64).reshape(self.n, A.NV)
        f = None if frame_idx is None else np.ascontiguousarray(frame_idx, dtype=np.int32)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        lib().emu_set_state(self.h, q.ctypes.data_as(A._dp), v.ctypes.data_as(A._dp),
                            None if f is None else f.ctypes.data_as(A._ip),
                            None if m is None else m.ctypes.data_as(C.POINTER(C.c_uint8)))

    def reset(self, mode=0, hard=1, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        lib().emu_reset(self.h, mode, hard, None if m is None else m.ctypes.data_as(C.POINTER(C.c_uint8)))

    def redo_total(self):
        """envs the packed path handed to the one-env kernel so far (a capacity of slot_kernel.h exceeded)"""
        return int(lib().emu_redo_total(self.h))

    def get_obs(self, out=None):
        q = self.get(A.F_QPOS); v = self.get(A.F_QVEL)
        o = np.concatenate([q[:, 7:], v[:, 6:]], 1)
        if out is not None:
            out[...] = o
            return out
        return o

    def close(self):
        pass

    def join(self):
        pass

    def sync(self):
        pass

    def debug_forward(self, env=0):
        buf = np.zeros(A.DEBUG_DOUBLES)
        lib().emu_debug_forward(self.h, env, buf.ctypes.data_as(A._dp))
        return A.parse_debug(buf)


def dispatch(n, first, count, nefc, solver_iter, arrival, order=None, with_tickets=True):
    """env_step.h order_ticket + dispatch_env on the testbench: the dispatch order of the launch after one in which the envs `arrival` (in that order)
    took their tickets.  Returns (one env per wave [count], four per wave [4 * ceil(count / 4)])."""
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    ne, it, ar = i32(nefc), i32(solver_iter), i32(arrival)
    od = None if order is None else i32(order)
    out1 = np.full(count, -1, np.int32); out4 = np.full(4 * ((count + 3) // 4), -1, np.int32)
    lib().emu_dispatch(n, first, count, ne.ctypes.data_as(A._ip), it.ctypes.data_as(A._ip), ar.ctypes.data_as(A._ip), len(ar),
                       None if od is None else od.ctypes.data_as(A._ip), 1 if with_tickets else 0, out1.ctypes.data_as(A._ip), out4.ctypes.data_as(A._ip))
    return out1, out4
