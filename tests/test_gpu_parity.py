"""GPU parity tests: the HIP path through the C ABI vs the CPU oracle (run with -m gpu on the MI355X box).
Tolerance: 1e-9 relative (north_star bar: 1e-5); contact (geom1, geom2) lists, row counts, PGS sweep counts and
done flags must be identical."""
import numpy as np
import pytest

from deepmimic_mujoco_amd import _abi as A
from tests import helpers as H

pytestmark = pytest.mark.gpu


def make_batch(n, flags=0, clip="walk"):
    from deepmimic_mujoco_amd import Batch
    mc = H.mocap(clip)
    return Batch(H.compiled_model(), mc.data_config, mc.data_vel, n, device=0, flags=flags, mocap_dt=float(mc.dt))


def test_native_library_loaded_and_device_visible():
    L = A.load()
    assert L.dm_device_count() >= 1


CLIPS = ["walk", "spinkick", "dance_b"]       # the clips of BASELINE.json configs[2], [3], [4]


@pytest.mark.parametrize("clip", CLIPS)
def test_forward_stages_match_oracle(clip):
    n = 48
    b = make_batch(n, clip=clip)
    worst = H.compare_forward(b, H.oracle_model(), *H.varied_states(n, seed=3, clip=clip))
    print("forward worst rel errs (%s):" % clip, {k: "%.1e" % v for k, v in worst.items()})
    b.close()


def test_capsule_box_contacts_match_oracle():
    qs = np.load(H.GOLDEN + "/capsule_box_poses.npy")
    n = len(qs)
    b = make_batch(n)
    H.compare_forward(b, H.oracle_model(), np.zeros(n, dtype=np.int32), qs, np.zeros((n, 34)), np.zeros((n, 34)), np.zeros((n, 28)))
    b.close()


def test_capsule_through_the_box_interior_matches_oracle():
    """Deep penetration: the capsule's axis passes through the inside of the box (zero-distance plateau; tests/test_wave_testbench.py)."""
    qs = np.load(H.GOLDEN + "/capsule_box_deep_poses.npy")
    n = len(qs)
    b = make_batch(n)
    H.compare_forward(b, H.oracle_model(), np.zeros(n, dtype=np.int32), qs, np.zeros((n, 34)), np.zeros((n, 34)), np.zeros((n, 28)))
    b.close()


def test_box_box_contacts_match_oracle():
    qs = np.load(H.GOLDEN + "/box_box_poses.npy")
    n = len(qs)
    b = make_batch(n)
    H.compare_forward(b, H.oracle_model(), np.zeros(n, dtype=np.int32), qs, np.zeros((n, 34)), np.zeros((n, 34)), np.zeros((n, 28)))
    idx = np.zeros(n, dtype=np.int32)
    H.compare_rollout(b, H.oracle_model(), idx, qs, np.zeros((n, 34)), steps=6, seed=9, action_scale=0.2)
    b.close()


def test_action_front_ends_match_host_formula():
    mc = H.mocap()
    n = 16
    idx, q, v, _w, _c = H.varied_states(n, seed=21)
    gold = np.load(H.GOLDEN + "/env_logic_golden.npz")
    rng = np.random.RandomState(3)
    for mode in (1, 2):
        b = make_batch(n)
        b.set_option(A.OPT_ACTION_MODE, mode)
        b.set_state(q, v, frame_idx=idx)
        a = rng.randn(n, 28) * 0.1
        b.step(a)
        dq = mc.data_config[idx][:, 7:] - q[:, 7:]
        expect = a + (0.8 * dq if mode == 1 else gold["kp"] * dq + gold["kd"] * (mc.data_vel[idx][:, 6:] - v[:, 6:]))
        assert np.abs(b.get(A.F_CTRL) - expect).max() < 1e-12
        b.close()


def test_wide_tier_takes_over_when_rows_exceed_32():
    """An env with more than 32 constraint rows is stepped by the 64-row kernel; results are identical to the oracle and to a
    batch forced onto the wide tier only."""
    mc = H.mocap(); cm = H.compiled_model()
    n = 8
    q = np.repeat(cm.qpos0[None], n, 0); v = np.zeros((n, 34))
    q[:, 2] = 0.9 - 0.018584 - 0.002                      # both feet flat, 2 mm into the floor: 8 corners x 4 = 32 rows
    q[:, 7] = 1.3; q[:, 16] = -0.1; q[:, 20] = -0.2       # + three violated limits -> 35 rows
    b = make_batch(n)
    worst, _ = H.compare_rollout(b, H.oracle_model(), np.zeros(n, dtype=np.int32), q, v, steps=3, seed=0, action_scale=0.1)
    b2 = make_batch(n); b2.set_option(102, 0)              # wide kernel only
    b3 = make_batch(n)
    for bb in (b2, b3):
        bb.set(A.F_QACC_WARMSTART, np.zeros((n, 34))); bb.set(A.F_TIME, np.zeros(n)); bb.set_state(q, v)
    rng = np.random.RandomState(0)
    most = 0
    for t in range(3):
        a = rng.randn(n, 28) * 0.1
        o2 = b2.step(a)[0].copy(); o3 = b3.step(a)[0].copy()
        most = max(most, int(b3.get(A.F_NEFC).max()))
        assert np.array_equal(o2, o3), "narrow+wide tiers must reproduce the wide-only result bit for bit"
    assert most > 32, "the pose was built to exceed the register tier (%d rows seen)" % most
    for bb in (b, b2, b3):
        bb.close()


@pytest.mark.parametrize("clip", CLIPS)
def test_rollout_matches_oracle_full_contact(clip):
    n = 32
    b = make_batch(n, clip=clip)
    idx, q, v, _ws, _c = H.varied_states(n, seed=5, clip=clip)
    worst, ndone = H.compare_rollout(b, H.oracle_model(), idx, q, v, steps=40 if clip == "walk" else 24, seed=1, clip=clip)
    print("rollout (%s) worst rel err %.2e, done events %d" % (clip, worst, ndone))
    b.close()


def test_rollout_no_contact_no_limit_config2():
    n = 16
    b = make_batch(n, flags=A.FLAG_NO_CONTACT | A.FLAG_NO_LIMIT)
    idx, q, v, _ws, _c = H.varied_states(n, seed=7)
    om = H.oracle_model(enable_contact=0, enable_limit=0)
    H.compare_rollout(b, om, idx, q, v, steps=30, seed=2)
    assert np.all(b.get(A.F_NEFC) == 0)
    b.close()


@pytest.mark.parametrize("clip", CLIPS)
@pytest.mark.parametrize("mode", [1, 2])
def test_reward_modes_match_oracle(mode, clip):
    n = 8
    b = make_batch(n, clip=clip)
    b.set_option(A.OPT_REWARD_MODE, mode)
    idx, q, v, _ws, _c = H.varied_states(n, seed=11, clip=clip)
    idx[0] = len(H.mocap(clip).data_config) - 2                      # one env's frame cursor wraps inside the test
    H.compare_rollout(b, H.oracle_model(), idx, q, v, steps=12, seed=3, reward_mode=mode, clip=clip)
    b.close()


def test_frame_skip_substeps():
    n = 4
    b = make_batch(n)
    idx, q, v, _ws, _c = H.varied_states(n, seed=13)
    H.compare_rollout(b, H.oracle_model(), idx, q, v, steps=5, seed=4, n_substeps=6)
    b.close()


def test_torch_device_pointers_and_determinism():
    import torch
    n = 64
    idx, q, v, _ws, _c = H.varied_states(n, seed=17)
    outs = []
    for rep in range(2):
        b = make_batch(n)
        b.set_state(q, v, frame_idx=idx)
        g = torch.Generator(device="cuda"); g.manual_seed(5)
        o = None
        for t in range(10):
            a = torch.randn((n, 28), generator=g, device="cuda", dtype=torch.float64) * 0.9
            o, r, d = b.step(a)
        torch.cuda.synchronize(); b.sync()
        outs.append((o.cpu().numpy().copy(), b.get(A.F_QPOS)))
        b.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]), "step is not bit-reproducible"


def test_reset_modes_and_autoreset():
    n = 256
    from deepmimic_mujoco_amd import DPVecEnv
    env = DPVecEnv(n, motion="walk", device=0, autoreset="rsi", seed=3)
    mc = env.mocap
    obs = env.reset("rsi")
    fi = env.batch.get(A.F_FRAME_IDX)
    assert fi.min() >= 0 and fi.max() < mc.data_config.shape[0] and len(np.unique(fi)) > 10
    assert np.array_equal(env.batch.get(A.F_QPOS), mc.data_config[fi])          # RSI copies the mocap frame bit-exactly
    assert np.array_equal(obs, np.concatenate([mc.data_config[fi][:, 7:], mc.data_vel[fi][:, 6:]], 1))
    obs = env.reset("init")
    q = env.batch.get(A.F_QPOS)
    assert np.all(np.abs(q - env._cm.qpos0) <= 0.01 + 1e-15) and np.abs(q - env._cm.qpos0).max() > 0.005
    assert np.all(env.batch.get(A.F_TIME) == 0) and np.all(env.batch.get(A.F_QACC_WARMSTART) == 0)
    # drive until some envs terminate; auto-reset must put them back on a mocap frame with time 0
    rng = np.random.RandomState(0)
    env.reset("rsi")
    seen = 0
    for t in range(60):
        obs, rew, done, _ = env.step(rng.randn(n, 28) * 0.9)
        dn = np.nonzero(done)[0]
        if len(dn):
            seen += len(dn)
            fi = env.batch.get(A.F_FRAME_IDX); q = env.batch.get(A.F_QPOS); tm = env.batch.get(A.F_TIME)
            assert np.array_equal(q[dn], mc.data_config[fi[dn]]) and np.all(tm[dn] == 0)
            assert np.array_equal(obs[dn][:, :28], mc.data_config[fi[dn]][:, 7:])
    assert seen > 0
    assert np.all(np.isfinite(env.batch.get(A.F_QPOS)))
    env.close()


def test_full_size_properties_4096():
    """BASELINE.json sizes, size-independent properties: free fall (no contacts/limits, zero ctrl) accelerates the
    COM at exactly -g; every env with identical inputs produces bit-identical outputs; state stays finite."""
    n = 4096
    b = make_batch(n, flags=A.FLAG_NO_CONTACT | A.FLAG_NO_LIMIT)
    idx, q1, v1, _ws, _c = H.varied_states(1, seed=23)
    q = np.repeat(q1, n, 0); v = np.repeat(v1, n, 0) * 0.0
    q[:, 2] += 5.0
    b.set_state(q, v)
    z0 = b.get(A.F_COM_Z)
    zs = [z0]
    for t in range(3):
        b.step(np.zeros((n, 28)))
        b.set_state(b.get(A.F_QPOS), b.get(A.F_QVEL))   # refresh derived data at the integrated state
        zs.append(b.get(A.F_COM_Z))
    h = 0.0166
    acc = (zs[2] - 2 * zs[1] + zs[0]) / h ** 2
    assert np.all(np.abs(acc + 9.81) < 1e-6), acc[:4]
    assert np.all(b.get(A.F_QPOS) == b.get(A.F_QPOS)[0])
    b.close()
    # full-contact batch from RSI: stays finite, contact lists well formed, no capacity overflow in the training regime
    b = make_batch(n)
    b.set_option(A.OPT_AUTORESET, 1)
    b.reset(0, 1)
    rng = np.random.RandomState(1)
    for t in range(20):
        b.step(rng.randn(n, 28) * 0.9)
    assert np.all(np.isfinite(b.get(A.F_QPOS))) and np.all(np.isfinite(b.get(A.F_QVEL)))
    cg = b.get(A.F_CONTACT_GEOMS); ncon = b.get(A.F_NCON)
    for e in range(0, n, 97):
        k = min(ncon[e], A.MAXEFC)
        assert np.all(cg[e][:k] >= 0) and np.all(cg[e][k:] == -1)
        assert np.all(cg[e][:k, 0] != cg[e][:k, 1])                 # a pair is two different geoms (geom1 = the lower geom TYPE, not id)
    assert (b.get(A.F_STATUS) & 1).mean() < 0.01
    b.close()


def test_bare_dpenv_runs_the_committed_default_clip_on_gpu():
    """`DPEnv()` with no arguments = the reference's committed configuration: Config.motion == 'dance_b' (src/config.py:9)."""
    import random
    from deepmimic_mujoco_amd import DPEnv
    from deepmimic_mujoco_amd.config import Config
    from oracle import oracle as O
    assert Config.motion == "dance_b"
    random.seed(4)
    env = DPEnv()
    mc = H.mocap("dance_b")
    assert env.mocap_data_len == len(mc.data_config) == 153 and abs(env.mocap_dt - 0.016667) < 1e-6
    random.seed(9); expect = random.randint(0, 152); random.seed(9)
    ob = env.reset()
    assert env.idx_init == expect and np.array_equal(ob, np.concatenate([mc.data_config[expect][7:], mc.data_vel[expect][6:]]))
    om = H.oracle_model(); od = O.Data(om); od.reset(); od.set_state(mc.data_config[expect], mc.data_vel[expect])
    rng = np.random.RandomState(0)
    for t in range(8):
        a = rng.randn(28) * 0.5
        ob, r, d, info = env.step(a)
        o, ro, do, _ = od.env_step(a)
        assert H.rel_err(ob, o) < 1e-9 and r == ro == 1.0 and d == do and info == {}
    env.close()


def test_dpenv_gym_surface_on_gpu():
    import random
    from deepmimic_mujoco_amd import DPEnv
    random.seed(0)
    env = DPEnv(motion="walk")
    assert env.action_space.shape == (28,) and env.observation_space.shape == (56,)
    ob = env.reset()
    assert ob.shape == (56,) and np.array_equal(ob[:28], env.mocap.data_config[env.idx_init][7:])
    ob2, r, d, info = env.step(env.action_space.sample())
    assert ob2.shape == (56,) and r == 1.0 and isinstance(d, bool) and info == {}
    ob3 = env.reset_model_init()
    assert np.all(np.abs(ob3) <= 0.01 + 1e-12)
    assert abs(env.get_time() - 0.0166) < 1e-12        # reset() zeroed time, one step since; reset_model_init keeps it
    env.close()


def test_overflow_strip_rows_match_oracle_and_the_all_register_tier():
    """More constraint rows than k_step_narrow keeps in registers: the remaining columns of A live in the per-env memory
    strip.  Must match the oracle, and the all-register kernel (option 102 = 0) to rounding."""
    idx, q, v = H.many_row_states(40, 64, want=6)
    n = len(q)
    b = make_batch(n)
    worst, _ = H.compare_rollout(b, H.oracle_model(), idx, q, v, steps=6, seed=4, action_scale=0.3)
    print("overflow-strip rollout worst rel err %.2e" % worst)
    b.close()
    outs = []
    for tier in (1, 0):
        bb = make_batch(n)
        bb.set_option(102, tier)
        bb.set(A.F_QACC_WARMSTART, np.zeros((n, 34))); bb.set_state(q, v, frame_idx=idx)
        rng = np.random.RandomState(0)
        o = [bb.step(rng.randn(n, 28) * 0.3)[0].copy() for _ in range(3)]
        assert bb.get(A.F_NEFC).max() > 40
        outs.append(np.stack(o)); bb.close()
    assert H.rel_err(outs[0], outs[1]) < 1e-11


def test_pgs_guarded_replay_path_gives_identical_results():
    """Option 103: every PGS sweep takes the guarded-replay path; must equal the speculative sweep bit for bit."""
    idx, q, v, _w, _c = H.varied_states(24, seed=17)
    i2, q2, v2 = H.many_row_states(32, 64, want=4)
    idx = np.concatenate([idx, i2]); q = np.concatenate([q, q2]); v = np.concatenate([v, v2])
    n = len(q)
    outs = []
    for force in (0, 1):
        b = make_batch(n)
        b.set_option(103, force)
        b.set(A.F_QACC_WARMSTART, np.zeros((n, 34))); b.set_state(q, v, frame_idx=idx)
        rng = np.random.RandomState(1)
        o = [b.step(rng.randn(n, 28) * 0.5)[0].copy() for _ in range(4)]
        outs.append((np.stack(o), b.get(A.F_SOLVER_ITER).copy(), b.get(A.F_QACC_WARMSTART).copy()))
        b.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][2], outs[1][2])


@pytest.mark.parametrize("n", [3000, 4096 + 37, 8192 + 101])      # > resident waves, so the reordering is active; <= 4096: sorted by the launch's last workgroup; > 8192: k_order's second key loop
def test_dispatch_order_does_not_change_results(n):
    """Longest-first dispatch (k_order, option 104) only changes which workgroup steps which env: outputs must be bit-identical."""
    outs = []
    for on in (1, 0):
        b = make_batch(n)
        b.set_option(104, on); b.set_option(A.OPT_AUTORESET, 1); b.set_option(A.OPT_SEED, 3)
        b.reset(0, 1)
        rng = np.random.RandomState(0)
        o = None
        for t in range(6):
            o, r, d = b.step(rng.randn(n, 28) * 0.9)
        outs.append((o.copy(), r.copy(), d.copy(), b.get(A.F_QPOS), b.get(A.F_NEFC)))
        b.close()
    for x, y in zip(outs[0], outs[1]):
        assert np.array_equal(x, y)
    assert outs[0][4].max() > 8                                        # contacts are present, so the order is not the identity


@pytest.mark.parametrize("clip,n,packed", [("spinkick", 4096, False), ("dance_b", 8192, False), ("dance_b", 8192, True)])     # one GPU's shard of BASELINE.json configs[3] / [4]; [4] also on the kernel DPVecEnv picks at that size
def test_full_size_shard_properties_cfg4_cfg5(clip, n, packed):
    """(`packed`: one or four environments per wavefront — the shard-cut invariance holds within either kernel; DPVecEnv's default picks by
    batch size, so a caller who compares differently sized batches bit for bit pins the choice.)
    Full per-GPU shard sizes of the 8-GPU configurations, size-independent properties: RSI + early termination + auto-reset
    from an interior shard's global env ids (shard 3: env_offset = 3 n); the step is bit-reproducible; results do not depend on
    how the shard is cut (two half batches with their own offsets reproduce the full batch bit for bit: per-env RNG streams
    are keyed by the GLOBAL env id); RSI resets land exactly on mocap frames; state stays finite; no capacity overflow."""
    import torch
    from deepmimic_mujoco_amd import DPVecEnv
    mc = H.mocap(clip); F = len(mc.data_config)
    steps = 12
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    acts = torch.randn((steps, n, 28), generator=g, device="cuda", dtype=torch.float64) * 0.9

    def run(lo, hi):
        m = hi - lo
        env = DPVecEnv(m, motion=clip, device=0, reward="imitation", autoreset="rsi", seed=11, env_offset=3 * n + lo, frame_skip=1, packed=packed)
        env.reset("rsi")
        fi0 = env.batch.get(A.F_FRAME_IDX).copy()
        dones = np.zeros(m, dtype=np.int64); rews = []
        for t in range(steps):
            obs, rew, done, _ = env.step(acts[t, lo:hi].contiguous())
            dones += done.cpu().numpy().astype(np.int64); rews.append(rew.cpu().numpy().copy())
        out = dict(obs=obs.cpu().numpy().copy(), rew=np.stack(rews), dones=dones, q=env.batch.get(A.F_QPOS), v=env.batch.get(A.F_QVEL),
                   fi=env.batch.get(A.F_FRAME_IDX), fi0=fi0, ep=env.batch.get(A.F_EPISODE), status=env.batch.get(A.F_STATUS),
                   nefc=env.batch.get(A.F_NEFC), t=env.batch.get(A.F_TIME))
        env.close()
        return out

    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", UserWarning)
        full = run(0, n); again = run(0, n)
        a, b = run(0, n // 2), run(n // 2, n)
    for k in ("obs", "rew", "q", "v", "fi", "ep", "dones"):
        assert np.array_equal(full[k], again[k]), "not bit-reproducible: " + k
        cat = np.concatenate([a[k], b[k]], axis=1 if k == "rew" else 0)
        assert np.array_equal(full[k], cat), "result depends on the sharding: " + k
    assert np.isfinite(full["q"]).all() and np.isfinite(full["v"]).all() and np.isfinite(full["rew"]).all()
    assert full["fi0"].min() >= 0 and full["fi0"].max() < F and len(np.unique(full["fi0"])) > min(F, 60) * 0.8
    for e in (0, 1, n // 3, n - 1):                                     # the device's draw = the host mirror of its counter-based stream
        assert full["fi0"][e] == H.device_rsi_frame(11, 3 * n + e, 0, F)
    assert full["dones"].sum() > 0, "early termination never fired in %d x %d steps" % (n, steps)
    fresh = np.nonzero(full["t"] == 0)[0]                                # envs reset by the last step: exactly on their mocap frame
    assert len(fresh) > 0 and np.array_equal(full["q"][fresh], mc.data_config[full["fi"][fresh]])
    assert (full["status"] & 1).mean() < 0.01 and full["nefc"].max() > 8
    assert 0 < full["rew"].min() and full["rew"].max() <= 1.0


@pytest.mark.parametrize("n", [1000, 4096 + 37])
def test_pipelined_sub_batches_do_not_change_results(n):
    """DM_OPT_PIPELINE cuts the env range into sub-batches stepped on their own streams, consecutive calls overlapping; it only
    changes WHEN an env is stepped: obs / reward / done of every step and the final state must be bit-identical for every depth,
    including a depth change in mid-run and host-pointer calls in between (which join)."""
    import torch
    steps = 24
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    acts = torch.randn((steps, n, 28), generator=g, device="cuda", dtype=torch.float64) * 0.9
    outs = []
    for depth in (1, 2, 3, 8):
        b = make_batch(n)
        b.set_option(A.OPT_AUTORESET, 1); b.set_option(A.OPT_SEED, 5); b.set_option(A.OPT_PIPELINE, depth)
        b.reset(0, 1)
        obs = torch.zeros((steps, n, 56), dtype=torch.float64, device="cuda"); rew = torch.zeros((steps, n), dtype=torch.float64, device="cuda")
        done = torch.zeros((steps, n), dtype=torch.uint8, device="cuda")
        for t in range(steps):
            if depth == 3 and t == 9:
                b.set_option(A.OPT_PIPELINE, 2)                        # depth change in mid-run
            b.step(acts[t], 1, (obs[t], rew[t], done[t]))              # no join between calls: consecutive steps overlap
        b.join()
        torch.cuda.current_stream().synchronize()
        outs.append((obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy(), b.get(A.F_QPOS), b.get(A.F_QACC_WARMSTART), b.get(A.F_EPISODE)))
        # a host-pointer step joins and runs as one launch: still the same trajectory as an unpipelined batch
        b.close()
    for o in outs[1:]:
        for x, y in zip(outs[0], o):
            assert np.array_equal(x, y)
    assert outs[0][2].sum() > 0 and outs[0][5].max() > 1, "no early termination / auto-reset inside the run"
    with pytest.raises(A.DmenvError):
        make_batch(8).set_option(A.OPT_PIPELINE, 9)


# ---- dtype 32: the float32 build of the same kernels (libdmenv32.so) -----------------------------------------
def test_float32_batch_tracks_the_float64_path():
    """Same C ABI, float32 arithmetic and device state.  Not a parity path (the bar of the float64 path is 1e-9 against the
    oracle); it must TRACK it: one step from identical states agrees to ~1e-4 of the observation scale wherever both paths see
    the same constraint rows, field reads / writes convert at the boundary, results are reproducible."""
    from deepmimic_mujoco_amd import Batch
    mc = H.mocap()
    n = 256
    idx, q, v, _ws, _c = H.varied_states(n, seed=31)
    q[:, 3:7] /= np.linalg.norm(q[:, 3:7], axis=1, keepdims=True)
    rng = np.random.RandomState(2)
    a = rng.randn(n, 28) * 0.5
    outs = {}
    for dt in (64, 32, 32):
        b = Batch(H.compiled_model(), mc.data_config, mc.data_vel, n, device=0, mocap_dt=float(mc.dt), dtype=dt)
        assert b.dtype == dt
        b.set(A.F_QACC_WARMSTART, np.zeros((n, 34))); b.set_state(q, v, frame_idx=idx)
        if dt == 32:                                          # boundary conversion: float64 in, float32 state, float64 out
            assert np.array_equal(b.get(A.F_QPOS), q.astype(np.float32).astype(np.float64))
        nefc0 = b.get(A.F_NEFC).copy()
        obs, rew, done = b.step(a)
        outs.setdefault(dt, []).append((obs.copy(), done.copy(), nefc0, b.get(A.F_NEFC).copy(), b.get(A.F_QPOS).copy()))
        b.close()
    o64, d64, n64a, n64b, q64 = outs[64][0]
    o32, d32, n32a, n32b, q32 = outs[32][0]
    assert np.array_equal(outs[32][0][0], outs[32][1][0]) and np.array_equal(q32, outs[32][1][4])          # reproducible
    same = (n64a == n32a) & (n64b == n32b)                    # same rows at the first and the last RK stage
    assert same.mean() > 0.9, same.mean()
    scale = np.maximum(1.0, np.abs(o64).max(1))
    err = np.abs(o32 - o64).max(1) / scale
    print("float32 vs float64 after one step: median rel err %.2e, 99th pct %.2e (envs with equal row counts: %.1f%%)"
          % (np.median(err[same]), np.percentile(err[same], 99), 100 * same.mean()))
    assert np.median(err[same]) < 2e-4 and np.percentile(err[same], 95) < 5e-3
    assert (d64 == d32).mean() > 0.99 and np.isfinite(o32).all()


def test_float32_batch_full_size_rollout_and_shipped_policy_anchor():
    """4096 float32 envs with RSI + early termination stay finite and keep auto-resetting exactly onto mocap frames (rounded to
    float32); the reference's shipped policy balances in the float32 physics as long as in the float64 one (~270 steps)."""
    import torch
    from deepmimic_mujoco_amd import DPVecEnv, MlpPolicy, traj_segment_generator
    from tests.test_policy import CKPT
    n = 4096
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", UserWarning)
        env = DPVecEnv(n, motion="walk", device=0, reward="imitation", autoreset="rsi", seed=5, dtype=32, frame_skip=1)
    mc = env.mocap
    env.reset("rsi")
    rng = np.random.RandomState(0)
    dones = 0
    for t in range(20):
        obs, rew, done, _ = env.step(rng.randn(n, 28) * 0.9)
        dones += int(done.sum())
    assert np.isfinite(obs).all() and np.isfinite(rew).all() and 0 < rew.min() and rew.max() <= 1.0 + 1e-6 and dones > 0
    fresh = np.nonzero(env.batch.get(A.F_TIME) == 0)[0]
    fi = env.batch.get(A.F_FRAME_IDX)
    assert len(fresh) > 0 and np.array_equal(env.batch.get(A.F_QPOS)[fresh], mc.data_config[fi[fresh]].astype(np.float32).astype(np.float64))
    env.close()
    lens = {}
    for dt in (64, 32):
        e = DPVecEnv(1024, motion="walk", device=0, reward="alive", autoreset="init", seed=1, dtype=dt)
        pol = MlpPolicy.from_tf_checkpoint(CKPT, device="cuda:0"); pol.seed(1)
        seg = next(traj_segment_generator(pol, e, 1200, stochastic=True, first_reset="init"))
        new = seg["new"].cpu().numpy()[1:]
        lens[dt] = float(np.where(new.any(0), new.argmax(0) + 1, 1200).mean())
        e.close()
    print("shipped policy, first-episode length: float64 %.1f, float32 %.1f" % (lens[64], lens[32]))
    assert abs(lens[32] / lens[64] - 1) < 0.15 and 200 < lens[32] < 360


# ---- four environments per wavefront (csrc/slot_kernel.h, DM option 105) ------------------------------------------------------------
@pytest.mark.parametrize("n", [13, 64])          # a last wave with spare slots; whole waves
def test_packed_kernel_matches_oracle_on_the_rowless_model_config2(n):
    """BASELINE.json configs[1] (contacts / limits off) through k_step_packed — four envs per wave, one 16-lane DPP row each — against
    the oracle, all three frame-indexed reward modes it covers; and against the one-env-per-wave kernel to rounding."""
    flags = A.FLAG_NO_CONTACT | A.FLAG_NO_LIMIT
    om = H.oracle_model(enable_contact=0, enable_limit=0)
    idx, q, v, _ws, _c = H.varied_states(n, seed=7)
    for mode in (0, 1, 2):
        b = make_batch(n, flags=flags)
        b.set_option(A.OPT_PACKED, 1); b.set_option(A.OPT_REWARD_MODE, mode)
        worst, nd = H.compare_rollout(b, om, idx, q, v, steps=12, seed=2, reward_mode=mode, n_substeps=2 if mode == 1 else 1)
        assert np.all(b.get(A.F_NEFC) == 0)
        b.close()
    outs = []
    for packed in (1, 0):
        b = make_batch(n, flags=flags)
        b.set_option(A.OPT_PACKED, packed); b.set_option(A.OPT_AUTORESET, 1); b.set_option(A.OPT_SEED, 3); b.set_option(A.OPT_ACTION_MODE, 1)
        b.reset(0, 1)
        rng = np.random.RandomState(0)
        o = [b.step(rng.randn(n, 28) * 0.3)[0].copy() for _ in range(8)]
        outs.append((np.stack(o), b.get(A.F_QPOS), b.get(A.F_EPISODE), b.get(A.F_XIPOS), b.get(A.F_COM_Z)))
        b.close()
    for x, y in zip(outs[0], outs[1]):
        assert H.rel_err(x, y) < 1e-11


@pytest.mark.parametrize("clip", CLIPS)
def test_packed_kernel_matches_oracle_full_contact(clip):
    """Contacts, joint limits, PGS on the four-envs-per-wave path against the oracle: 64 varied states (several exceed the packed path's
    capacities and come back through the one-env code), 30 steps, all frame-indexed reward modes; sweep counts, row counts and contact
    lists of the last evaluation identical."""
    from oracle import oracle as O
    n = 64
    om = H.oracle_model()
    idx, q, v, _ws, _c = H.varied_states(n, seed=5, clip=clip)
    for mode in (0, 1, 2):
        b = make_batch(n, clip=clip)
        b.set_option(A.OPT_PACKED, 1); b.set_option(A.OPT_REWARD_MODE, mode)
        worst, nd = H.compare_rollout(b, om, idx, q, v, steps=30 if mode == 0 else 8, seed=1, reward_mode=mode, clip=clip)
        if mode == 0:
            print("packed rollout (%s): worst rel err %.2e, %d done, redo %s" % (clip, worst, nd, b.redo_reasons()))
            assert b.redo_total() > 0
        b.close()
    # diagnostics of a packed step: PGS sweep counts, row / contact counts, contact geom lists
    b = make_batch(n, clip=clip); b.set_option(A.OPT_PACKED, 1)
    b.set(A.F_QACC_WARMSTART, np.zeros((n, 34))); b.set_state(q, v, frame_idx=idx)
    ods = [O.Data(om) for _ in range(n)]
    for e in range(n):
        ods[e].reset(); ods[e].set_state(q[e], v[e])
    rng = np.random.RandomState(0)
    for t in range(4):
        a = rng.randn(n, 28) * 0.9
        b.step(a)
        for e in range(n):
            ods[e].env_step(a[e])
        assert np.array_equal(b.get(A.F_SOLVER_ITER), np.array([int(d.get("solver_iter")[0]) for d in ods]))
        assert np.array_equal(b.get(A.F_NEFC), np.array([int(d.get("nefc")[0]) for d in ods]))
        onc = np.array([int(d.get("ncon")[0]) for d in ods])
        assert np.array_equal(b.get(A.F_NCON), onc)
        cg = b.get(A.F_CONTACT_GEOMS)
        for e in range(n):
            k = min(int(onc[e]), A.MAXEFC)
            assert np.array_equal(cg[e][:k], ods[e].get("contact_geom").reshape(-1, 2).astype(np.int32)[:k]) and np.all(cg[e][k:] == -1)
    b.close()


@pytest.mark.parametrize("packed", [False, True])
def test_self_ordering_launches_step_every_env_exactly_once_whatever_runs_in_between(packed):
    """Per-step launches order themselves (env_step.h dispatch_env / order_ticket; dmenv.hip ord_bind): three rotating ticket phases per pipelined
    part.  Device-pointer steps of 4 096 + 37 envs (more than the resident waves, so the order is active) through every hand-over of the host's
    bookkeeping — one launch per step, two and three pipelined parts, a partition change, a horizon launch, a reset and a host-pointer step in
    between: after every step each env's clock has advanced by exactly one time step (an env stepped twice or not at all shows there), and the
    whole run is bit-identical to the same run with the dispatch order off (option 104)."""
    import torch
    n, dt = 4096 + 37, 0.0166
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(5)
    T = 26
    ac = torch.randn((T, n, 28), generator=g, dtype=torch.float64, device=dev) * 0.9
    finals = []
    for on in (1, 0):
        b = make_batch(n)
        b.set_option(A.OPT_PACKED, 1 if packed else 0); b.set_option(104, on); b.set_option(A.OPT_AUTORESET, 1); b.set_option(A.OPT_SEED, 9)
        b.set_option(A.OPT_DIAGNOSTICS, 0)
        b.reset(0, 1)
        ob = torch.zeros((n, 56), dtype=torch.float64, device=dev); rw = torch.zeros(n, dtype=torch.float64, device=dev); dn = torch.zeros(n, dtype=torch.uint8, device=dev)
        t_prev = b.get(A.F_TIME).copy(); ep_prev = b.get(A.F_EPISODE).copy()
        hist = []
        for t in range(T):
            if t == 4: b.set_option(A.OPT_PIPELINE, 2)
            if t == 10: b.set_option(A.OPT_PIPELINE, 3)          # a new partition: the old parts' tickets must not be used
            if t == 14: b.set_option(A.OPT_PIPELINE, 1)
            if t == 18: b.set_option(A.OPT_PIPELINE, 2)
            if t == 16:                                          # a horizon launch in between (packed only): tickets grow stale, stay a permutation
                if packed:
                    o2 = torch.zeros((3, n, 56), dtype=torch.float64, device=dev); r2 = torch.zeros((3, n), dtype=torch.float64, device=dev); d2 = torch.zeros((3, n), dtype=torch.uint8, device=dev)
                    b.set_option(106, 1); b.rollout(ac[:4].contiguous(), (o2, r2, d2), 1); b.set_option(106, -1)
                    b.join(); b.sync()
                    hist.append(o2.cpu().numpy().copy())
                    t_prev = b.get(A.F_TIME).copy(); ep_prev = b.get(A.F_EPISODE).copy()
            if t == 20:                                          # a host-pointer step (one launch over the whole batch while two parts are configured)
                o, r, d = b.step(ac[t].cpu().numpy())
                hist.append(o.copy())
            else:
                b.step(ac[t], 1, (ob, rw, dn)); b.join(); b.sync()
                hist.append(ob.cpu().numpy().copy())
            tm = b.get(A.F_TIME); ep = b.get(A.F_EPISODE)
            fresh = ep != ep_prev                                # auto-reset: the clock restarts
            assert np.allclose(np.where(fresh, dt, tm - t_prev), dt, rtol=0, atol=1e-12), "step %d: an env was stepped twice or not at all" % t
            t_prev, ep_prev = tm.copy(), ep.copy()
            if t == 7: b.reset(0, 1); t_prev = b.get(A.F_TIME).copy(); ep_prev = b.get(A.F_EPISODE).copy()
        finals.append((hist, b.get(A.F_QPOS), b.get(A.F_QVEL), b.get(A.F_NEFC)))
        b.close()
    for x, y in zip(finals[0][0], finals[1][0]):
        assert np.array_equal(x, y)
    for i in (1, 2, 3):
        assert np.array_equal(finals[0][i], finals[1][i])
    assert finals[0][3].max() > 8
