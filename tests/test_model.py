"""Model compiler (spec -> constant tables), MJCF reader, and their agreement with the oracle's own compile."""
import os

import numpy as np
import pytest

from deepmimic_mujoco_amd.humanoid import humanoid_spec
from deepmimic_mujoco_amd.mjcf import load_mjcf, to_mjcf
from deepmimic_mujoco_amd.model import CompiledModel
from tests import helpers as H

REF_XML = "/root/reference/src/mujoco/humanoid_deepmimic/envs/asset/dp_env_v3.xml"


def _same(a, b, path=""):
    if isinstance(a, dict):
        for k in a:
            if k != "name":
                _same(a[k], b[k], path + "/" + k)
    elif isinstance(a, (list, tuple)) and a and isinstance(a[0], (dict, list, tuple)):
        assert len(a) == len(b), path
        for i, (x, y) in enumerate(zip(a, b)):
            _same(x, y, "%s[%d]" % (path, i))
    elif isinstance(a, (list, tuple)):
        assert tuple(float(x) for x in a) == tuple(float(x) for x in b), path
    else:
        assert a == b, (path, a, b)


def test_sizes_and_known_answers():
    cm = H.compiled_model()
    assert (cm.nq, cm.nv, cm.nu, cm.nbody, cm.ngeom, cm.njnt) == (35, 34, 28, 14, 16, 29)
    assert abs(cm.total_mass - 45.0) < 1e-12
    assert abs(cm.init_com_z - 0.91075) < 1e-5                     # hand-derived COM height at qpos0
    assert abs(cm.body_ipos[5][2] + 0.166316) < 1e-6               # elbow = capsule 1.0 kg + wrist 0.5 kg
    assert cm.npair == 104 and np.all(cm.pair_geom[:15, 0] == 0)   # 15 floor pairs first, then 89 body-body pairs
    assert list(cm.actuator_gear[:6]) == [200, 200, 200, 50, 50, 50] and np.all(cm.actuator_ctrlrange == [-0.5, 0.5])
    assert np.array_equal(cm.qpos0[:7], [0, 0, 0.9, 1, 0, 0, 0])
    M = cm.mass_matrix(cm.qpos0)
    assert np.allclose(M, M.T) and np.linalg.eigvalsh(M).min() > 0


def test_agrees_with_oracle_compile():
    from oracle import oracle as O
    cm = H.compiled_model(); om = O.Model()
    for f, a in [("body_mass", cm.body_mass), ("body_ipos", cm.body_ipos), ("body_inertia", cm.body_inertia),
                 ("body_invweight0", cm.body_invweight0), ("dof_invweight0", cm.dof_invweight0), ("qpos0", cm.qpos0),
                 ("geom_quat", cm.geom_quat), ("geom_lpos", cm.geom_pos), ("geom_lsize", cm.geom_size),
                 ("dof_parent", cm.dof_parentid), ("pair_g1", cm.pair_geom[:, 0]), ("pair_g2", cm.pair_geom[:, 1])]:
        assert np.abs(om.get(f) - np.asarray(a, dtype=np.float64).ravel()).max() < 1e-13, f
    assert abs(om.get("meaninertia")[0] - cm.meaninertia) < 1e-13
    rng = np.random.RandomState(0)
    d = O.Data(om)
    for _ in range(3):
        q = cm.qpos0.copy(); q[7:] = rng.uniform(-1, 1, 28); q[3:7] = rng.randn(4); q[:3] += rng.randn(3)
        d.set("qpos", q); d.forward()
        assert np.abs(d.get("M").reshape(34, 34) - cm.mass_matrix(q)).max() < 1e-12   # CRB (oracle) vs Jacobian sum (host)


def test_mjcf_roundtrip_and_rejects_unsupported():
    spec = humanoid_spec()
    _same(spec, load_mjcf(to_mjcf(spec)))
    with pytest.raises(ValueError):
        load_mjcf('<mujoco><worldbody><body><joint type="ball"/></body></worldbody></mujoco>')
    with pytest.raises(ValueError):
        load_mjcf('<mujoco><worldbody/><equality/></mujoco>')


@pytest.mark.skipif(not os.path.exists(REF_XML), reason="reference checkout not present")
def test_builtin_table_equals_reference_xml():
    _same(humanoid_spec(), load_mjcf(REF_XML))


def test_rig_variant_compiles_with_other_options():
    spec = humanoid_spec()
    spec["option"]["timestep"] = 0.005
    spec["joints"][5]["range"] = (-0.3, 0.4)
    cm = CompiledModel(spec)
    assert cm.timestep == 0.005 and tuple(cm.jnt_range[5]) == (-0.3, 0.4)
    spec["option"]["integrator"] = "Euler"
    with pytest.raises(ValueError):
        CompiledModel(spec)
