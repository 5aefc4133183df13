"""ctypes mirror of include/dmenv.h (structures, constants, prototypes) and the libdmenv.so loader.

The product path has no CPU fallback: `load()` raises if libdmenv.so is missing or cannot be loaded, and
`dm_batch_create` fails with DM_ENODEVICE when no HIP device is visible.
"""
import ctypes as C
import os

import numpy as np

ABI_VERSION = 4
NBODY, NJNT, NQ, NV, NU, NGEOM, NOBS, MAXEFC = 14, 29, 35, 34, 28, 16, 56, 64
DEBUG_DOUBLES = 34 * 34 + 34 * 3 + 42 + 3 + MAXEFC * (34 + 6)
PTR_HOST, PTR_DEVICE = 0, 1
FLAG_NO_CONTACT, FLAG_NO_LIMIT = 1, 2
OPT_REWARD_MODE, OPT_AUTORESET, OPT_ACTION_MODE, OPT_SEED, OPT_DIAGNOSTICS, OPT_PIPELINE, OPT_PACKED, OPT_ENV_OFFSET = 1, 2, 3, 4, 5, 6, 7, 100
OPT_STEP_QUEUE = 8
MAX_PIPELINE = 8
MAX_STEP_QUEUE = 256
# per-environment capacities of the OPT_PACKED path (include/dmenv.h DM_PACKED_*)
PACKED_MAXROWS, PACKED_MAXLIMROWS, PACKED_MAXCON, PACKED_MAXFRAME, PACKED_MAXCAND = 40, 16, 13, 8, 32      # (rows: inside a horizon launch)
PACKED_MAXROWS_PER_STEP = 32
(F_QPOS, F_QVEL, F_QACC_WARMSTART, F_TIME, F_FRAME_IDX, F_FRAME_INIT, F_XIPOS, F_COM_Z, F_NCON, F_NEFC,
 F_CONTACT_GEOMS, F_STATUS, F_SOLVER_ITER, F_CTRL, F_EPISODE, F_CYCLE) = range(1, 17)

# field -> (numpy dtype, per-env shape)
FIELD_SPEC = {
    F_QPOS: (np.float64, (NQ,)), F_QVEL: (np.float64, (NV,)), F_QACC_WARMSTART: (np.float64, (NV,)),
    F_TIME: (np.float64, ()), F_FRAME_IDX: (np.int32, ()), F_FRAME_INIT: (np.int32, ()),
    F_XIPOS: (np.float64, (NBODY, 3)), F_COM_Z: (np.float64, ()), F_NCON: (np.int32, ()), F_NEFC: (np.int32, ()),
    F_CONTACT_GEOMS: (np.int32, (MAXEFC, 2)), F_STATUS: (np.int32, ()), F_SOLVER_ITER: (np.int32, ()),
    F_CTRL: (np.float64, (NU,)), F_EPISODE: (np.int32, ()), F_CYCLE: (np.int32, ()),
}

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


class ModelDesc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("nbody", C.c_int32), ("njnt", C.c_int32), ("nq", C.c_int32), ("nv", C.c_int32), ("nu", C.c_int32),
        ("ngeom", C.c_int32), ("npair", C.c_int32), ("iterations", C.c_int32),
        ("body_parentid", _ip), ("body_dofnum", _ip), ("body_pos", _dp), ("body_ipos", _dp), ("body_mass", _dp),
        ("body_inertia", _dp), ("body_invweight0", _dp),
        ("jnt_type", _ip), ("jnt_bodyid", _ip), ("jnt_limited", _ip), ("jnt_axis", _dp), ("jnt_range", _dp),
        ("dof_armature", _dp), ("dof_damping", _dp), ("dof_invweight0", _dp),
        ("geom_type", _ip), ("geom_bodyid", _ip), ("geom_condim", _ip), ("geom_pos", _dp), ("geom_mat", _dp),
        ("geom_size", _dp), ("geom_margin", _dp), ("geom_friction", _dp),
        ("pair_geom", _ip),
        ("actuator_dofid", _ip), ("actuator_gear", _dp), ("actuator_ctrlrange", _dp),
        ("timestep", C.c_double), ("gravity", C.c_double * 3), ("tolerance", C.c_double),
        ("solref", C.c_double * 2), ("solimp", C.c_double * 5), ("meaninertia", C.c_double),
    ]


def make_model_desc(cm):
    """Fill a ModelDesc from a `model.CompiledModel`; returns (desc, keepalive list of arrays)."""
    keep = []

    def d(a):
        a = np.ascontiguousarray(a, dtype=np.float64); keep.append(a); return a.ctypes.data_as(_dp)

    def i(a):
        a = np.ascontiguousarray(a, dtype=np.int32); keep.append(a); return a.ctypes.data_as(_ip)

    md = ModelDesc()
    md.abi_version = ABI_VERSION
    md.nbody, md.njnt, md.nq, md.nv, md.nu, md.ngeom = cm.nbody, cm.njnt, cm.nq, cm.nv, cm.nu, cm.ngeom
    md.npair, md.iterations = cm.npair, cm.iterations
    md.body_parentid = i(cm.body_parentid); md.body_dofnum = i(cm.body_dofnum)
    md.body_pos = d(cm.body_pos); md.body_ipos = d(cm.body_ipos); md.body_mass = d(cm.body_mass)
    md.body_inertia = d(cm.body_inertia); md.body_invweight0 = d(cm.body_invweight0)
    md.jnt_type = i(cm.jnt_type); md.jnt_bodyid = i(cm.jnt_bodyid); md.jnt_limited = i(cm.jnt_limited)
    md.jnt_axis = d(cm.jnt_axis); md.jnt_range = d(cm.jnt_range)
    md.dof_armature = d(cm.dof_armature); md.dof_damping = d(cm.dof_damping); md.dof_invweight0 = d(cm.dof_invweight0)
    md.geom_type = i(cm.geom_type); md.geom_bodyid = i(cm.geom_bodyid); md.geom_condim = i(cm.geom_condim)
    md.geom_pos = d(cm.geom_pos); md.geom_mat = d(cm.geom_mat); md.geom_size = d(cm.geom_size)
    md.geom_margin = d(cm.geom_margin); md.geom_friction = d(cm.geom_friction)
    md.pair_geom = i(cm.pair_geom)
    md.actuator_dofid = i(cm.actuator_dofid); md.actuator_gear = d(cm.actuator_gear)
    md.actuator_ctrlrange = d(cm.actuator_ctrlrange)
    md.timestep, md.tolerance, md.meaninertia = cm.timestep, cm.tolerance, cm.meaninertia
    for k in range(3):
        md.gravity[k] = cm.gravity[k]
    for k in range(2):
        md.solref[k] = cm.solref[k]
    for k in range(5):
        md.solimp[k] = cm.solimp[k]
    return md, keep


_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DMENV_LIB") or os.path.join(_HERE, "csrc", "libdmenv.so")   # DMENV_LIB: developer override (A/B builds)
EXPORTS = ["dm_model_create", "dm_model_destroy", "dm_mocap_create", "dm_mocap_set_imitation", "dm_mocap_destroy", "dm_batch_create",
           "dm_batch_destroy", "dm_batch_set_stream", "dm_batch_set_option", "dm_batch_set_state", "dm_batch_reset",
           "dm_batch_step", "dm_batch_get_obs", "dm_batch_get", "dm_batch_set", "dm_batch_debug_forward",
           "dm_batch_last_step_ms", "dm_batch_enable_timing", "dm_batch_read_profile", "dm_batch_sync", "dm_batch_join", "dm_policy_weight_count", "dm_policy_act", "dm_batch_step_act", "dm_batch_rollout", "dm_vf_param_count", "dm_vf_scratch_bytes", "dm_vf_fit_epoch", "dm_pg_param_count", "dm_pg_scratch_bytes", "dm_pg_losses", "dm_pg_fvp", "dm_batch_redo_total", "dm_batch_queue_stats", "dm_gae", "dm_episode_scan", "dm_rms_scratch_bytes", "dm_rms_update", "dm_last_error", "dm_abi_version", "dm_real_bits",
           "dm_device_count"]
_LIB = None


class DmenvError(RuntimeError):
    pass


def load(dtype=64):
    """Load libdmenv.so (float64 arithmetic; dtype=32: libdmenv32.so, the float32 build of the same source), built by
    `__graft_entry__.build()` / csrc/build.py.  Both export the same C ABI (float64 buffers).  No fallback of any kind."""
    global _LIB
    if dtype not in (64, 32):
        raise ValueError("dtype must be 64 or 32")
    if _LIB is None:
        _LIB = {}
    if dtype in _LIB:
        return _LIB[dtype]
    path = LIB_PATH if dtype == 64 else LIB_PATH[:-3] + "32.so"
    if not os.path.exists(path):
        raise DmenvError("%s not found at %s — build the HIP extension first "
                         "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback" % (os.path.basename(path), path))
    L = C.CDLL(path)
    vp, i32, u8p = C.c_void_p, C.c_int32, C.POINTER(C.c_uint8)
    L.dm_last_error.restype = C.c_char_p
    L.dm_model_create.argtypes = [C.POINTER(ModelDesc), C.POINTER(vp)]
    L.dm_model_destroy.argtypes = [vp]; L.dm_model_destroy.restype = None
    L.dm_mocap_create.argtypes = [_dp, _dp, i32, C.c_double, C.POINTER(vp)]
    L.dm_mocap_set_imitation.argtypes = [vp, _dp, i32, _dp]
    L.dm_mocap_destroy.argtypes = [vp]; L.dm_mocap_destroy.restype = None
    L.dm_batch_create.argtypes = [vp, vp, i32, i32, C.c_uint32, C.POINTER(vp)]
    L.dm_batch_destroy.argtypes = [vp]; L.dm_batch_destroy.restype = None
    L.dm_batch_set_stream.argtypes = [vp, vp]
    L.dm_batch_set_option.argtypes = [vp, i32, C.c_int64]
    L.dm_batch_set_state.argtypes = [vp, vp, vp, vp, vp, i32]
    L.dm_batch_reset.argtypes = [vp, i32, i32, vp, i32]
    L.dm_batch_step.argtypes = [vp, vp, vp, vp, vp, i32, i32]
    L.dm_batch_get_obs.argtypes = [vp, vp, i32]
    L.dm_batch_get.argtypes = [vp, i32, vp, C.c_size_t, i32]
    L.dm_batch_set.argtypes = [vp, i32, vp, C.c_size_t, i32]
    L.dm_batch_debug_forward.argtypes = [vp, i32, _dp]
    L.dm_batch_last_step_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.dm_batch_enable_timing.argtypes = [vp, i32]
    L.dm_batch_read_profile.argtypes = [vp, C.POINTER(C.c_longlong)]
    L.dm_batch_sync.argtypes = [vp]
    L.dm_batch_join.argtypes = [vp]
    L.dm_policy_act.argtypes = [vp, vp, vp, vp, i32, i32, C.c_uint64, C.c_uint64, vp]
    L.dm_vf_scratch_bytes.argtypes = [i32, i32]; L.dm_vf_scratch_bytes.restype = C.c_size_t
    L.dm_batch_redo_total.argtypes = [vp, C.POINTER(C.c_int64)]
    L.dm_batch_queue_stats.argtypes = [vp, C.POINTER(C.c_int64)]
    L.dm_pg_scratch_bytes.argtypes = []; L.dm_pg_scratch_bytes.restype = C.c_size_t
    L.dm_pg_losses.argtypes = [vp, i32, vp, vp, vp, vp, i32, vp, vp, vp, C.c_double, i32, vp, vp, vp, vp, i32]
    L.dm_pg_fvp.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, i32]
    L.dm_vf_fit_epoch.argtypes = [vp, vp, i32, i32, vp, vp, vp, C.POINTER(C.c_float), C.c_double, C.c_double, C.c_double, vp, vp, vp, vp, vp, vp, vp, i32]
    L.dm_batch_step_act.argtypes = [vp, vp, vp, vp, vp, i32, vp, vp, vp, i32, C.c_uint64, C.c_uint64]
    L.dm_batch_rollout.argtypes = [vp, vp, vp, vp, vp, i32, i32, vp, vp, i32, C.c_uint64, C.c_uint64]
    L.dm_gae.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, C.c_double, C.c_double, vp]
    L.dm_rms_scratch_bytes.restype = C.c_size_t
    L.dm_rms_update.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp]
    L.dm_episode_scan.argtypes = [vp, vp, i32, i32, vp, vp, vp, i32, vp, vp]
    if L.dm_abi_version() != ABI_VERSION:
        raise DmenvError("libdmenv.so ABI version %d != %d" % (L.dm_abi_version(), ABI_VERSION))
    if L.dm_real_bits() != dtype:
        raise DmenvError("%s computes in float%d, expected float%d" % (os.path.basename(path), L.dm_real_bits(), dtype))
    _LIB[dtype] = L
    return L


def check(rc, L=None):
    if rc != 0:
        L = L or load()
        raise DmenvError("libdmenv error %d: %s" % (rc, (L.dm_last_error() or b"").decode()))


def parse_debug(buf):
    """Split the DM_DEBUG_DOUBLES dump of dm_batch_debug_forward into named arrays."""
    o = 0
    out = {}
    out["M"] = buf[o:o + 34 * 34].reshape(34, 34); o += 34 * 34
    out["qfrc_bias"] = buf[o:o + 34]; o += 34
    out["qacc_smooth"] = buf[o:o + 34]; o += 34
    out["qacc"] = buf[o:o + 34]; o += 34
    out["xipos"] = buf[o:o + 42].reshape(14, 3); o += 42
    out["nefc"], out["ncon"], out["solver_iter"] = int(buf[o]), int(buf[o + 1]), int(buf[o + 2]); o += 3
    rows = buf[o:o + MAXEFC * 40].reshape(MAXEFC, 40)
    n = out["nefc"]
    out["efc_J"] = rows[:n, :34]
    for k, name in enumerate(["efc_pos", "efc_margin", "efc_R", "efc_aref", "efc_b", "efc_force"]):
        out[name] = rows[:n, 34 + k]
    return out
