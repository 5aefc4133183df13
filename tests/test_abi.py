"""The C-ABI library: loads, exports every symbol include/dmenv.h declares, validates its inputs on the host, and
refuses to run without a HIP device (no CPU fallback).  No compute calls here (no GPU in the build container)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from deepmimic_mujoco_amd import _abi as A
from deepmimic_mujoco_amd.humanoid import humanoid_spec
from deepmimic_mujoco_amd.model import CompiledModel
from tests import helpers as H

HEADER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include", "dmenv.h")


@pytest.mark.parametrize("dtype", [64, 32])
def test_library_exports_every_declared_symbol(dtype):
    """libdmenv.so (float64 arithmetic) and libdmenv32.so (the float32 build of the same source) export the same C ABI."""
    L = A.load(dtype)
    text = open(HEADER).read()
    declared = sorted(set(re.findall(r"\b(dm_[a-z_0-9]+)\s*\(", text)))
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(L, name), "the float%d library does not export %s" % (dtype, name)
    assert sorted(declared) == sorted(A.EXPORTS)
    assert L.dm_abi_version() == A.ABI_VERSION and L.dm_real_bits() == dtype


def test_header_constants_match_python_mirror():
    text = open(HEADER).read()
    for name, val in [("DM_NQ", A.NQ), ("DM_NV", A.NV), ("DM_NU", A.NU), ("DM_NOBS", A.NOBS), ("DM_NBODY", A.NBODY),
                      ("DM_MAXEFC", A.MAXEFC)]:
        assert int(re.search(r"#define %s (\d+)" % name, text).group(1)) == val
    assert A.DEBUG_DOUBLES == 34 * 34 + 34 * 3 + 42 + 3 + 64 * 40
    for k, name in [(A.F_QPOS, "DM_F_QPOS"), (A.F_EPISODE, "DM_F_EPISODE"), (A.F_CONTACT_GEOMS, "DM_F_CONTACT_GEOMS")]:
        assert int(re.search(r"%s = (\d+)" % name, text).group(1)) == k


def test_model_create_validates_topology_and_reports_errors():
    L = A.load()
    md, keep = A.make_model_desc(H.compiled_model())
    h = C.c_void_p()
    assert L.dm_model_create(C.byref(md), C.byref(h)) == 0 and h.value
    L.dm_model_destroy(h)
    assert L.dm_model_create(None, C.byref(h)) < 0 and b"null" in L.dm_last_error()
    md2, keep2 = A.make_model_desc(H.compiled_model())
    md2.abi_version = 99
    assert L.dm_model_create(C.byref(md2), C.byref(h)) < 0 and b"ABI" in L.dm_last_error()
    # a different tree must be rejected, not silently simulated
    spec = humanoid_spec()
    spec["bodies"][3]["parent"] = 1      # neck hangs off the root instead of the chest
    md3, keep3 = A.make_model_desc(CompiledModel(spec))
    rc = L.dm_model_create(C.byref(md3), C.byref(h))
    assert rc == -4 and b"topology" in L.dm_last_error()


def test_mocap_create_rejects_bad_input():
    L = A.load()
    h = C.c_void_p()
    assert L.dm_mocap_create(None, None, 0, 0.0, C.byref(h)) < 0
    mc = H.mocap()
    cfg = np.ascontiguousarray(mc.data_config); vel = np.ascontiguousarray(mc.data_vel)
    assert L.dm_mocap_create(cfg.ctypes.data_as(A._dp), vel.ctypes.data_as(A._dp), cfg.shape[0], float(mc.dt), C.byref(h)) == 0
    L.dm_mocap_destroy(h)


def test_no_cpu_fallback_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    from deepmimic_mujoco_amd import Batch, DPEnv
    mc = H.mocap()
    with pytest.raises(A.DmenvError) as ei:
        Batch(H.compiled_model(), mc.data_config, mc.data_vel, 4)
    assert "no HIP device" in str(ei.value) or "-5" in str(ei.value)
    with pytest.raises(A.DmenvError):
        DPEnv(motion="walk")


def test_product_package_never_imports_the_oracle():
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "deepmimic_mujoco_amd")
    for dirpath, _d, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("the CPU oracle", "").replace("CPU oracle", "").replace("the oracle", "") or f in ("wave.h",), f
