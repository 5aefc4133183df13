"""5-term imitation reward (spec: code.md:1017-1143).  Three implementations of one feature
definition — numpy (deepmimic_mujoco_amd/imitation.py, also builds the reference table), the C oracle, the HIP epilogue —
must agree; known answers: reward == 1 exactly on the reference motion, each term reacts to its own perturbation only."""
import numpy as np
import pytest

from deepmimic_mujoco_amd import _abi as A
from deepmimic_mujoco_amd.imitation import ImitationSpec, FEAT, TERM_W, TERM_SCALE, O_EE, O_COMV, O_JQ
from tests import helpers as H


def _spec():
    return ImitationSpec(H.compiled_model())


def test_reward_is_one_on_the_reference_and_weights_are_normalised():
    sp = _spec(); mc = H.mocap()
    T = sp.build_table(mc.data_config, mc.data_vel)
    assert T.shape == (len(mc.data_config), FEAT)
    assert abs(sp.w_joint.sum() + sp.w_root - 1) < 1e-12 and abs(TERM_W.sum() - 1) < 1e-12     # code.md:1025-1031
    for k in (0, 7, 20, len(T) - 1):
        f = sp.features(mc.data_config[k], mc.data_vel[k])
        assert np.array_equal(f, T[k]) and abs(sp.reward(f, T[k]) - 1) < 1e-12
    # a shifted reference root (completed cycles) is matched by shifting the simulated root
    q = mc.data_config[3].copy(); q[0:2] += [2.5, -0.4]
    assert abs(sp.reward(sp.features(q, mc.data_vel[3]), T[3], root_shift=(2.5, -0.4)) - 1) < 1e-12


def test_each_term_reacts_to_its_own_perturbation():
    sp = _spec(); mc = H.mocap()
    k = 11
    q0, v0 = mc.data_config[k], mc.data_vel[k]
    ref = sp.features(q0, v0)
    z = np.zeros(34)
    q = q0.copy(); q[7 + 14] += 0.3                                     # one right-hip hinge (at rest): pose + that foot's end effector
    e = sp.reward_terms(sp.features(q, z), sp.features(q0, z))
    assert e[0] > 0 and e[1] == 0 and e[2] > 0 and e[3] == 0 and abs(e[4]) < 1e-12
    qa = q0.copy(); qa[7 + 20] += 0.3                                   # an ankle hinge turns about the end effector itself
    ea = sp.reward_terms(sp.features(qa, v0), ref)
    w_ankle = sp.w_joint[list(sp.cm.body_names).index("right_ankle") - 2]
    assert ea[2] < 1e-20 and abs(ea[0] - w_ankle * sp_theta2(sp, q0, qa)) < 1e-12
    v = v0.copy(); v[6 + 3] += 0.5                                      # one neck hinge rate: velocity (+ a little COM velocity)
    e = sp.reward_terms(sp.features(q0, v), ref)
    assert e[0] == 0 and e[1] > 0 and e[2] == 0 and e[3] == 0 and 0 < e[4] < 1e-3
    q = q0.copy(); q[2] += 0.05                                         # root height: root term and end-effector heights
    e = sp.reward_terms(sp.features(q, v0), ref)
    assert e[0] == 0 and abs(e[3] - 0.05 ** 2) < 1e-12 and abs(e[2] - 0.05 ** 2) < 1e-12 and abs(e[4]) < 1e-12
    v = v0.copy(); v[0] += 1.0                                          # root linear velocity: root (0.01 |dv|^2) and COM velocity
    e = sp.reward_terms(sp.features(q0, v), ref)
    assert abs(e[3] - 0.01) < 1e-12 and abs(e[4] - 0.1) < 1e-9 and e[0] == 0 and e[2] == 0
    # 1-hinge joints use the squared angle difference; 3-hinge joints the squared rotation angle of the difference quaternion
    q = q0.copy(); q[7 + 17] += 0.2                                     # right knee
    e = sp.reward_terms(sp.features(q, v0), ref)
    w_knee = sp.w_joint[list(sp.cm.body_names).index("right_knee") - 2]
    assert abs(e[0] - w_knee * 0.2 ** 2) < 1e-12
    assert w_ankle > 0 and abs(float((TERM_W * np.exp(-TERM_SCALE * e)).sum()) - sp.reward(sp.features(q, v0), ref)) < 1e-15


def sp_theta2(sp, q0, q1):
    from deepmimic_mujoco_amd.imitation import quat_diff_theta
    g = list(sp.cm.body_names).index("right_ankle") - 2
    f0, f1 = sp.features(q0, np.zeros(34)), sp.features(q1, np.zeros(34))
    return quat_diff_theta(f0[O_JQ + 4 * g:O_JQ + 4 * g + 4], f1[O_JQ + 4 * g:O_JQ + 4 * g + 4]) ** 2


def test_heading_invariance_of_the_end_effector_term():
    sp = _spec(); mc = H.mocap()
    q0, v0 = mc.data_config[9].copy(), mc.data_vel[9]
    f0 = sp.features(q0, v0)
    yaw = 1.1
    qz = np.array([np.cos(yaw / 2), 0, 0, np.sin(yaw / 2)])
    from deepmimic_mujoco_amd.imitation import quat_mul
    q1 = q0.copy(); q1[3:7] = quat_mul(qz, q0[3:7])
    f1 = sp.features(q1, v0)
    assert np.allclose(f0[O_EE:O_EE + 12], f1[O_EE:O_EE + 12], atol=1e-12)     # end effectors live in the heading frame
    assert np.allclose(f0[O_JQ:O_JQ + 48], f1[O_JQ:O_JQ + 48], atol=1e-15)


def test_oracle_features_and_reward_match_the_numpy_definition():
    from oracle import oracle as O
    sp = _spec(); mc = H.mocap()
    om = H.oracle_model()
    params = sp.params(mc.data_config, mc.loop)
    T = sp.build_table(mc.data_config, mc.data_vel)
    rng = np.random.RandomState(0)
    idx, q, v, _w, _c = H.varied_states(12, seed=2)
    for e in range(12):
        f_np = sp.features(q[e], v[e])
        f_c = O.imitation_features(om, q[e], v[e], params)
        assert np.abs(f_np - f_c).max() < 1e-12 * max(1.0, np.abs(f_np).max()), (e, np.abs(f_np - f_c).argmax())
        k = int(idx[e]); sh = rng.randn(2)
        r_c, t_c = O.imitation_reward(om, f_c, T[k], params, shift=sh)
        assert abs(r_c - sp.reward(f_np, T[k], root_shift=sh)) < 1e-12 and np.allclose(t_c, sp.reward_terms(f_np, T[k], sh), rtol=1e-10, atol=1e-13)


def test_oracle_imitation_episode_wraps_with_cycle_shift_and_ends_non_looping_clips():
    from oracle import oracle as O
    sp = _spec(); mc = H.mocap()
    om = H.oracle_model(); d = O.Data(om)
    params = sp.params(mc.data_config, mc.loop)
    T = sp.build_table(mc.data_config, mc.data_vel)
    F = len(T)
    d.reset(); d.set_state(mc.data_config[F - 3], mc.data_vel[F - 3])
    idx, cyc = F - 3, 0
    seen = []
    for t in range(4):
        o, r, dn, idx, cyc = O.env_step_imitation(om, d, np.zeros(28), 2, T, params, idx, cyc)
        seen.append((idx, cyc)); assert 0 < r <= 1
    assert seen == [(F - 2, 0), (F - 1, 0), (0, 1), (1, 1)]
    p2 = params.copy(); p2[15] = 0                                       # "Loop: none": hold the last frame, episode ends
    d.reset(); d.set_state(mc.data_config[F - 2], mc.data_vel[F - 2])
    o, r, dn, idx, cyc = O.env_step_imitation(om, d, np.zeros(28), 2, T, p2, F - 2, 0)
    assert (idx, cyc, dn) == (F - 1, 0, False)
    o, r, dn, idx, cyc = O.env_step_imitation(om, d, np.zeros(28), 2, T, p2, idx, cyc)
    assert (idx, cyc, dn) == (F - 1, 0, True)
