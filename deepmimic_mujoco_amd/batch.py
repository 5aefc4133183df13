"""`Batch`: thin Python owner of a `dm_batch` (include/dmenv.h) — N lock-step environments on one GPU.

Buffers may be numpy arrays (host pointers; the library stages them) or torch tensors on the batch's device
(device pointers, zero-copy, stream-ordered).  Nothing here computes physics: every call lands in a HIP kernel.
"""
import ctypes as C

import numpy as np

from . import _abi as A


def _is_torch(x):
    return type(x).__module__.startswith("torch")


class Batch(object):
    def __init__(self, compiled_model, data_config, data_vel, n_envs, device=0, flags=0, mocap_dt=0.0, imitation=None, dtype=64):
        """dtype: arithmetic / device-state type of the kernels, 64 (default: the parity path) or 32 (the float32 build of the same
        source, libdmenv32.so: faster, ~1e-4 relative per step against the float64 path).  Buffers are float64 either way."""
        L = A.load(dtype)
        self.dtype = int(dtype)
        self._L = L
        self.n = int(n_envs)
        self.device = int(device)
        md, self._keep = A.make_model_desc(compiled_model)
        self._model = C.c_void_p(); self._mocap = C.c_void_p(); self._h = C.c_void_p()
        A.check(L.dm_model_create(C.byref(md), C.byref(self._model)), L)
        cfg = np.ascontiguousarray(data_config, dtype=np.float64); vel = np.ascontiguousarray(data_vel, dtype=np.float64)
        if cfg.ndim != 2 or cfg.shape[1] != A.NQ or vel.shape != (cfg.shape[0], A.NV):
            raise ValueError("mocap tables must be [F,35] and [F,34]")
        self.n_frames = cfg.shape[0]
        A.check(L.dm_mocap_create(cfg.ctypes.data_as(A._dp), vel.ctypes.data_as(A._dp), cfg.shape[0], float(mocap_dt),
                                  C.byref(self._mocap)), L)
        if imitation is not None:            # (table [F,112], params [32]) from imitation.ImitationSpec: enables reward mode 3
            tab = np.ascontiguousarray(imitation[0], dtype=np.float64); par = np.ascontiguousarray(imitation[1], dtype=np.float64)
            if tab.shape != (cfg.shape[0], 112) or par.shape != (32,):
                raise ValueError("imitation = (table [F,112], params [32])")
            A.check(L.dm_mocap_set_imitation(self._mocap, tab.ctypes.data_as(A._dp), 112, par.ctypes.data_as(A._dp)), L)
        A.check(L.dm_batch_create(self._model, self._mocap, self.n, self.device, int(flags), C.byref(self._h)), L)

    def close(self):
        L = getattr(self, "_L", None)
        if L is None:
            return
        if getattr(self, "_h", None):
            L.dm_batch_destroy(self._h); self._h = None
        if getattr(self, "_mocap", None):
            L.dm_mocap_destroy(self._mocap); self._mocap = None
        if getattr(self, "_model", None):
            L.dm_model_destroy(self._model); self._model = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- helpers ------------------------------------------------------------------------------------
    def _ptr(self, x, dtype, shape, out=False):
        """-> (pointer, kind, keepalive)"""
        if x is None:
            return None, A.PTR_HOST, None
        if _is_torch(x):
            import torch
            want = {np.float64: torch.float64, np.int32: torch.int32, np.uint8: torch.uint8}[dtype]
            if x.dtype != want or not x.is_contiguous() or tuple(x.shape) != tuple(shape):
                raise ValueError("tensor must be contiguous %s of shape %s" % (want, shape))
            if x.device.type != "cuda" or (x.device.index or 0) != self.device:
                raise ValueError("tensor must live on cuda:%d" % self.device)
            # device buffers are ordered on the batch's stream: follow torch's current stream so that kernels producing
            # the tensor (e.g. the policy, torch.randn) and our launch are serialised without extra synchronisation
            cur = torch.cuda.current_stream(x.device).cuda_stream
            if cur != getattr(self, "_stream_handle", None):
                self.set_stream(cur)
            return C.c_void_p(x.data_ptr()), A.PTR_DEVICE, x
        a = x if out else np.ascontiguousarray(x, dtype=dtype)
        if a.dtype != dtype or not a.flags.c_contiguous or a.shape != tuple(shape):
            raise ValueError("array must be C-contiguous %s of shape %s" % (np.dtype(dtype), shape))
        return C.c_void_p(a.ctypes.data), A.PTR_HOST, a

    _queue_refs = None

    def set_option(self, opt, value):
        A.check(self._L.dm_batch_set_option(self._h, int(opt), int(value)), self._L)
        self.__dict__.setdefault("options", {})[int(opt)] = int(value)
        if int(opt) == A.OPT_STEP_QUEUE:
            self._queue_refs = [] if int(value) > 0 else None

    def queue_stats(self):
        """OPT_STEP_QUEUE: (horizon launches issued for queued steps, steps they carried, steps queued now)"""
        v = (C.c_int64 * 3)()
        A.check(self._L.dm_batch_queue_stats(self._h, v), self._L)
        return int(v[0]), int(v[1]), int(v[2])

    @property
    def can_step_act(self):
        """dm_batch_step_act needs the two-tier kernel (option 102, default on) and no per-stage profiling (option 101)."""
        o = self.__dict__.get("options", {})
        return o.get(102, 1) != 0 and o.get(101, 0) == 0

    def set_stream(self, stream_handle):
        A.check(self._L.dm_batch_set_stream(self._h, C.c_void_p(int(stream_handle))), self._L)
        self._stream_handle = int(stream_handle)

    # ---- kernel choice by workload (DPVecEnv(packed=None)) ---------------------------------------------------------------
    ADAPT_EVERY = 256            # steps between looks at the batch's row statistics (one small D2H + stream sync each)
    REDO_RATE_MAX = 3e-4         # packed -> one-env: env-steps per env-step that overflowed the packed path's capacities
    HEAVY_ROWS = 30              # one-env -> packed: no environment of the batch holds more constraint rows than this (per-step packed launches hold 32)
    HEAVY_ROWS_EXT = 38          # ... -> packed with the three-set code (OPT_PACKED 2: 40 rows per env)

    def enable_auto_packed(self, on=True):
        """Let the batch choose between one and four environments per wavefront (DM_OPT_PACKED) from what it is simulating: the packed
        kernel is 1.4-1.5x faster while environments stay within its per-env capacities (_abi.PACKED_*: 32 rows per step, 13 contacts), but every
        environment beyond them is re-stepped one per wave AFTER the packed launch.  A population that stands on both feet (33 .. 37 rows for a few per
        cent of its env-steps) is moved to the per-step launches with the three-set code (OPT_PACKED 2: 40 rows per env, ~8 % slower otherwise) and back
        when nobody holds more than 32 rows; overflows of another kind send the batch to the one-env kernel.  Every ADAPT_EVERY steps the redo rate and
        its reasons (packed) or the largest row count (one-env) of the batch decide; the choice is a deterministic function of the trajectory."""
        self._auto = bool(on)
        self._auto_ctr = 0
        self._redo_last = self.redo_total() if on else 0
        self._redo_rows_last = self.redo_reasons()[4] if on else 0
        self._calm_checks = 0
        self.auto_switches = 0

    CALM_CHECKS = 3              # OPT_PACKED 2 -> 1 only after this many consecutive looks with nobody above 32 rows (a population near the redo threshold
                                 #  holds ~2.5 such environments per step: a single snapshot reads zero one time in twelve and the batch would flip 2 -> 1 -> 2)

    def rebaseline_auto(self):
        """The redo counters moved while the chooser was suspended (SegmentCollector's horizon launches re-step inside the wave and count there): start
        the next window from where they stand now."""
        r = self.redo_reasons()
        self._redo_last = r[0]; self._redo_rows_last = r[4]
        self._calm_checks = 0

    def _adapt(self):
        self._auto_ctr += 1
        if self._auto_ctr % self.ADAPT_EVERY:
            return
        mode = int(self.__dict__.get("options", {}).get(A.OPT_PACKED, 0))
        if mode:
            reasons = self.redo_reasons()
            redo = reasons[0]
            rate = (redo - self._redo_last) / float(self.ADAPT_EVERY * self.n)
            rows_share = (reasons[4] - self.__dict__.get("_redo_rows_last", 0)) / float(max(1, redo - self._redo_last))
            self._redo_last = redo; self._redo_rows_last = reasons[4]
            if rate > self.REDO_RATE_MAX:
                self._calm_checks = 0
                # too many environments beyond the per-step capacities.  Nearly all of them for ROWS (a population standing on both feet: 33 .. 37): the
                # per-step launches with the three-set code (OPT_PACKED 2, 40 rows per env); otherwise — or if that was already on — the one-env kernel
                self.set_option(A.OPT_PACKED, 2 if (mode == 1 and rows_share >= 0.9) else 0); self.auto_switches += 1
            elif mode == 2:
                # nobody above 32 rows at CALM_CHECKS looks in a row: the lean per-step kernel (8 % faster)
                self._calm_checks = self._calm_checks + 1 if int((self.get(A.F_NEFC) > A.PACKED_MAXROWS_PER_STEP).sum()) == 0 else 0
                if self._calm_checks >= self.CALM_CHECKS:
                    self._calm_checks = 0
                    self.set_option(A.OPT_PACKED, 1); self.auto_switches += 1
        else:
            top = int(self.get(A.F_NEFC).max())
            if top <= self.HEAVY_ROWS_EXT:
                self.set_option(A.OPT_PACKED, 1 if top <= self.HEAVY_ROWS else 2); self.auto_switches += 1
                self.rebaseline_auto()

    # ---- the hot path -------------------------------------------------------------------------------
    def _step_device(self, action, n_substeps, out):
        """`step` for device tensors with every buffer given: the same checks as `_ptr`, written out once (the generic path costs ~10 us
        of Python per call — 4 % of a 250 us step).  Returns None when the arguments are not of that form."""
        obs, rew, done = out
        try:
            dev = action.device
            if not (action.is_cuda and obs.is_cuda and rew.is_cuda and done.is_cuda):
                return None
        except AttributeError:
            return None
        import torch
        n = self.n
        if not (action.dtype == obs.dtype == rew.dtype == torch.float64 and done.dtype == torch.uint8 and action.is_contiguous() and obs.is_contiguous()
                and rew.is_contiguous() and done.is_contiguous() and action.shape == (n, A.NU) and obs.shape == (n, A.NOBS) and rew.shape == (n,) and done.shape == (n,)):
            return None                                            # the generic path raises the precise error
        if (dev.index or 0) != self.device or obs.device != dev or rew.device != dev or done.device != dev:
            return None
        cur = torch.cuda.current_stream(dev).cuda_stream
        if cur != getattr(self, "_stream_handle", None):
            self.set_stream(cur)
        rc = self._L.dm_batch_step(self._h, action.data_ptr(), obs.data_ptr(), rew.data_ptr(), done.data_ptr(), int(n_substeps), A.PTR_DEVICE)
        if rc:
            A.check(rc, self._L)
        if self._queue_refs is not None:
            self._queue_refs.append((action, obs, rew, done))
            if len(self._queue_refs) > 2 * A.MAX_STEP_QUEUE:
                del self._queue_refs[:A.MAX_STEP_QUEUE]
        return out

    def step(self, action, n_substeps=1, out=None):
        if self.__dict__.get("_auto"):
            self._adapt()
        if out is not None:
            r = self._step_device(action, n_substeps, out)
            if r is not None:
                return r
        n = self.n
        ap, kind, _ka = self._ptr(action, np.float64, (n, A.NU))
        if out is None:
            if kind == A.PTR_DEVICE:
                import torch
                dev = action.device
                out = (torch.empty((n, A.NOBS), dtype=torch.float64, device=dev), torch.empty(n, dtype=torch.float64, device=dev),
                       torch.empty(n, dtype=torch.uint8, device=dev))
            else:
                out = (np.empty((n, A.NOBS)), np.empty(n), np.empty(n, dtype=np.uint8))
        obs, rew, done = out
        op, k1, _ = self._ptr(obs, np.float64, (n, A.NOBS), out=True)
        rp, k2, _ = self._ptr(rew, np.float64, (n,), out=True)
        dp, k3, _ = self._ptr(done, np.uint8, (n,), out=True)
        if not (kind == k1 == k2 == k3):
            raise ValueError("action and output buffers must all be host arrays or all be device tensors")
        A.check(self._L.dm_batch_step(self._h, ap, op, rp, dp, int(n_substeps), kind), self._L)
        if self._queue_refs is not None and kind == A.PTR_DEVICE:
            # OPT_STEP_QUEUE: the call may only have been queued — its tensors must outlive the flush (any other entry point of the batch)
            self._queue_refs.append((action, obs, rew, done))
            if len(self._queue_refs) > 2 * A.MAX_STEP_QUEUE:
                del self._queue_refs[:A.MAX_STEP_QUEUE]      # (older than any step that can still be queued)
        return obs, rew, done

    def step_act(self, action, n_substeps, out, weights, next_action, next_vpred, stochastic, seed, counter):
        """`step` followed, inside the step kernel, by the policy's step on the new observations (dm_batch_step_act): device tensors
        only.  action [n, 28] f64 -> out = (obs [n, 56] f64, rew [n] f64, done [n] u8); next_action [n, 28] f64 and next_vpred [n] f32
        receive the action / value for those observations; `weights` is MlpPolicy.pack()'s float32 block."""
        if self.__dict__.get("_auto"):
            self._adapt()
        n = self.n
        ap, kind, _ka = self._ptr(action, np.float64, (n, A.NU))
        obs, rew, done = out
        op, k1, _ = self._ptr(obs, np.float64, (n, A.NOBS), out=True)
        rp, k2, _ = self._ptr(rew, np.float64, (n,), out=True)
        dp, k3, _ = self._ptr(done, np.uint8, (n,), out=True)
        np_, k4, _ = self._ptr(next_action, np.float64, (n, A.NU), out=True)
        if not (kind == k1 == k2 == k3 == k4 == A.PTR_DEVICE):
            raise ValueError("step_act works on device tensors")
        import torch
        if not (weights.is_cuda and weights.dtype == torch.float32 and weights.is_contiguous() and weights.numel() == self._L.dm_policy_weight_count()):
            raise ValueError("weights: the packed float32 policy block (MlpPolicy.pack()) on the device")
        if not (next_vpred.is_cuda and next_vpred.dtype == torch.float32 and next_vpred.is_contiguous() and next_vpred.numel() == n):
            raise ValueError("next_vpred: float32 [n] on the device")
        A.check(self._L.dm_batch_step_act(self._h, ap, op, rp, dp, int(n_substeps), C.c_void_p(weights.data_ptr()), np_,
                                          C.c_void_p(next_vpred.data_ptr()), 1 if stochastic else 0, int(seed) & (2 ** 64 - 1), int(counter)), self._L)
        return obs, rew, done

    def rollout(self, action, out, n_substeps=1, weights=None, vpred=None, stochastic=True, seed=0, counter=0):
        """T steps in one call (dm_batch_rollout), device tensors only: action [T + 1, n, 28] f64 (row t feeds step t; with `weights` — the
        packed policy block — rows 1..T are written by the policy), out = (obs [T, n, 56] f64, rew [T, n] f64, done [T, n] u8), vpred [T, n]
        f32 (with weights).  Same results as T `step` / `step_act` calls; on the packed path (option 105) one launch for the horizon."""
        import torch
        obs, rew, done = out
        T, n = int(obs.shape[0]), self.n
        ap, k0, _ = self._ptr(action, np.float64, (T + 1, n, A.NU), out=True)
        op, k1, _ = self._ptr(obs, np.float64, (T, n, A.NOBS), out=True)
        rp, k2, _ = self._ptr(rew, np.float64, (T, n), out=True)
        dp, k3, _ = self._ptr(done, np.uint8, (T, n), out=True)
        if not (k0 == k1 == k2 == k3 == A.PTR_DEVICE):
            raise ValueError("rollout works on device tensors")
        wp = vp = None
        if weights is not None:
            if not (weights.is_cuda and weights.dtype == torch.float32 and weights.is_contiguous() and weights.numel() == self._L.dm_policy_weight_count()):
                raise ValueError("weights: the packed float32 policy block (MlpPolicy.pack()) on the device")
            if vpred is None or not (vpred.is_cuda and vpred.dtype == torch.float32 and vpred.is_contiguous() and tuple(vpred.shape) == (T, n)):
                raise ValueError("vpred: float32 [T, n] on the device")
            wp, vp = C.c_void_p(weights.data_ptr()), C.c_void_p(vpred.data_ptr())
        A.check(self._L.dm_batch_rollout(self._h, ap, op, rp, dp, T, int(n_substeps), wp, vp, 1 if stochastic else 0, int(seed) & (2 ** 64 - 1), int(counter)), self._L)
        return obs, rew, done

    def get_obs(self, out=None):
        if out is None:
            out = np.empty((self.n, A.NOBS))
        p, kind, _ = self._ptr(out, np.float64, (self.n, A.NOBS), out=True)
        A.check(self._L.dm_batch_get_obs(self._h, p, kind), self._L)
        return out

    def set_state(self, qpos, qvel, frame_idx=None, mask=None):
        n = self.n
        qp, k, _a = self._ptr(qpos, np.float64, (n, A.NQ)); vp, k2, _b = self._ptr(qvel, np.float64, (n, A.NV))
        fp, k3, _c = self._ptr(frame_idx, np.int32, (n,)); mp, k4, _d = self._ptr(mask, np.uint8, (n,))
        kinds = {k, k2} | ({k3} if frame_idx is not None else set()) | ({k4} if mask is not None else set())
        if len(kinds) != 1:
            raise ValueError("mixed host/device buffers")
        A.check(self._L.dm_batch_set_state(self._h, qp, vp, fp, mp, k), self._L)

    def reset(self, mode=0, hard=1, mask=None):
        mp, k, _ = self._ptr(mask, np.uint8, (self.n,))
        A.check(self._L.dm_batch_reset(self._h, int(mode), int(hard), mp, k), self._L)

    def get(self, field, out=None):
        dt, shp = A.FIELD_SPEC[field]
        if out is None:
            out = np.empty((self.n,) + shp, dtype=dt)
        p, kind, _ = self._ptr(out, dt, (self.n,) + shp, out=True)
        nbytes = int(np.prod((self.n,) + shp)) * np.dtype(dt).itemsize
        A.check(self._L.dm_batch_get(self._h, int(field), p, nbytes, kind), self._L)
        return out

    def set(self, field, value):
        dt, shp = A.FIELD_SPEC[field]
        if not _is_torch(value):
            value = np.ascontiguousarray(np.asarray(value, dtype=dt).reshape((self.n,) + shp))
        p, kind, _ka = self._ptr(value, dt, (self.n,) + shp)
        nbytes = int(np.prod((self.n,) + shp)) * np.dtype(dt).itemsize
        A.check(self._L.dm_batch_set(self._h, int(field), p, nbytes, kind), self._L)

    def debug_forward(self, env=0):
        buf = np.zeros(A.DEBUG_DOUBLES)
        A.check(self._L.dm_batch_debug_forward(self._h, int(env), buf.ctypes.data_as(A._dp)), self._L)
        return A.parse_debug(buf)

    def enable_timing(self, on=True):
        A.check(self._L.dm_batch_enable_timing(self._h, 1 if on else 0), self._L)

    def last_step_ms(self):
        ms = C.c_float(0)
        A.check(self._L.dm_batch_last_step_ms(self._h, C.byref(ms)), self._L)
        return float(ms.value)

    def read_profile(self):
        """[N,32] int64 shader-clock cycles of the last step: kin, mass, bias, rows, constraint, total, nefc, sweeps | 8..13 the
        constraint stage's parts | 16..23 the collision stage's parts, 24..27 its near-pair counts (include/dmenv.h)."""
        out = np.zeros((self.n, 32), dtype=np.int64)
        A.check(self._L.dm_batch_read_profile(self._h, out.ctypes.data_as(C.POINTER(C.c_longlong))), self._L)
        return out

    def redo_total(self):
        """env-steps the four-envs-per-wave path (option 105) handed to the one-env kernel so far"""
        return self.redo_reasons()[0]

    def redo_reasons(self):
        """[total, then by reason: > PACKED_MAXCAND pairs past the bounding spheres, box staging slots, > PACKED_MAXCON contacts (or > PACKED_MAXFRAME
        contact pairs), > PACKED_MAXROWS rows (or > PACKED_MAXLIMROWS limits), PGS cost test] env-steps handed to the one-env code (_abi.PACKED_*)"""
        v = (C.c_int64 * 8)()
        A.check(self._L.dm_batch_redo_total(self._h, v), self._L)
        return [int(x) for x in v[:6]]

    def sync(self):
        A.check(self._L.dm_batch_sync(self._h), self._L)
        if self._queue_refs:
            del self._queue_refs[:]

    def join(self):
        """Run the steps OPT_STEP_QUEUE has queued and make the batch's stream wait for every pipelined step launch still in flight
        (OPT_PIPELINE > 1), without a host wait.  Call before consuming the outputs of queued / pipelined `step` calls on that stream."""
        A.check(self._L.dm_batch_join(self._h), self._L)
