"""Full-size oracle parity (run with -m gpu): the REAL per-GPU shard of BASELINE.json configs[2], [3], [4] — 4 096 'walk', 4 096
'spinkick', 8 192 'dance_b' environments — stepped exactly as bench.py steps it (one dm_batch_step per step on either kernel, or the whole
run as ONE dm_batch_rollout launch) (5-term imitation reward, RSI auto-reset from the
device's counter-based RNG, two pipelined sub-batches, longest-first dispatch order, an interior shard's global env ids) against the
CPU oracle's OpenMP batch step, EVERY env, every step: observations and rewards to 1e-9, done flags, frame cursors, cycle counters,
constraint-row counts, contact counts and contact (geom1, geom2) lists identical.  The oracle side mirrors the device's auto-reset
on the host (tests/helpers.device_rsi_frame: the frame `reset_env` draws for (seed, global env id, episode)).
Round 4 adds: the step-queue form (DM_OPT_STEP_QUEUE: the per-step calls queued and run as one horizon launch), 'spinkick' on the per-step
packed kernel, a 64-step horizon (second-episode RSI draws and clip wraps for most environments), BASELINE.json configs[1] (contacts and
limits off, the P-controller of src/env_torque_test.py:14-20) at its full 4 096 envs on the kernels bench.py times it on, and a STANDING
population (the shipped policy's regime: 32 rows per env most of the time).
Round 5 adds: the judged launch of bench.py itself (256 queued steps, one k_rollout_packed launch: every env of all 256 steps), the three-set per-step
kernels ('packed-ext'), every env's contact list in the standing forms.
Round 6 adds: the two sharded configs at longer horizons — 'spinkick' x 4 096 as one 256-step queued launch, 'dance_b' x 8 192 as one 64-step horizon launch
(second- and later-episode RSI draws, wraps of the 78- / 153-frame clips) — and qacc_warmstart held to 1e-8.
Reference semantics: src/dp_env_v3.py:106-156 (step, is_done, reset_model)."""
import os

import numpy as np
import pytest

from deepmimic_mujoco_amd import _abi as A
from tests import helpers as H

pytestmark = pytest.mark.gpu

SEED = 11
STEPS = 16


@pytest.mark.parametrize("clip,n,packed", [("walk", 4096, 0), ("spinkick", 4096, 0), ("dance_b", 8192, 0), ("dance_b", 8192, 1), ("walk", 4096, 1), ("walk", 4096, 2), ("spinkick", 4096, 2), ("dance_b", 8192, 2),
                                           ("spinkick", 4096, 1), ("walk", 4096, 3), ("dance_b", 8192, 3), ("walk", 4096, 64), ("walk", 4096, 256),
                                           ("spinkick", 4096, 256), ("dance_b", 8192, 64)])
def test_full_shard_matches_oracle_every_env_every_step(clip, n, packed):
    import torch
    STEPS = 16
    queue = packed == 3                              # 3: four per wave, the 16 dm_batch_step calls QUEUED (DM_OPT_STEP_QUEUE) and run as one horizon launch at the join
    if packed == 64:                                 # 64: one 64-step horizon launch — most environments see a second episode's RSI draw, 'walk' (39 frames) wraps
        STEPS, packed = 64, 2
    if packed == 256:                                # 256: THE JUDGED LAUNCH of bench.py — 256 dm_batch_step calls queued (DM_OPT_STEP_QUEUE = 256) and run as one
        STEPS, packed, queue = 256, 1, True          #      k_rollout_packed launch at the join: every env of every one of its 256 steps against the oracle
    from deepmimic_mujoco_amd import Batch
    from deepmimic_mujoco_amd.imitation import ImitationSpec
    from oracle import oracle as O
    mc = H.mocap(clip)
    sp = ImitationSpec(H.compiled_model())
    T, P = sp.table_for(mc)
    F = len(T)
    off = 3 * n                                                   # shard 3 of 8: global env ids do not start at 0
    b = Batch(H.compiled_model(), mc.data_config, mc.data_vel, n, device=0, mocap_dt=float(mc.dt), imitation=(T, P))
    b.set_option(A.OPT_REWARD_MODE, 3); b.set_option(A.OPT_AUTORESET, 1); b.set_option(A.OPT_SEED, SEED)
    b.set_option(A.OPT_ENV_OFFSET, off); b.set_option(A.OPT_DIAGNOSTICS, 1); b.set_option(A.OPT_PIPELINE, 2)
    b.set_option(A.OPT_PACKED, 1 if packed else 0)   # 0: one environment per wavefront; 1: four (what DPVecEnv picks from 4 096 envs up);
    horizon = packed == 2                            # 2: four, and all 16 steps in ONE launch (dm_batch_rollout: every wave at its own pace)
    if queue:
        b.set_option(A.OPT_STEP_QUEUE, max(64, STEPS))
    b.reset(0, 1)                                                 # env.reset(): sim.reset() + RSI
    fidx = b.get(A.F_FRAME_IDX).copy()
    expect0 = np.array([H.device_rsi_frame(SEED, off + e, 0, F) for e in range(n)], dtype=np.int32)
    assert np.array_equal(fidx, expect0), "device RSI draw differs from the host mirror of its RNG"
    om = H.oracle_model()
    ods = [O.Data(om) for _ in range(n)]
    for e in range(n):
        ods[e].reset(); ods[e].set_state(mc.data_config[fidx[e]], mc.data_vel[fidx[e]])
    cyc = np.zeros(n, dtype=np.int32)
    episode = np.ones(n, dtype=np.int64)                          # the initial reset consumed episode 0
    nthreads = max(1, len(os.sched_getaffinity(0)))
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    dev = torch.device("cuda:0")
    acts = torch.empty((STEPS + 1, n, 28), dtype=torch.float64, device=dev)
    for t in range(STEPS):
        acts[t] = torch.randn((n, 28), generator=g, device=dev, dtype=torch.float64) * 0.9
    obs_T = torch.empty((STEPS, n, 56), dtype=torch.float64, device=dev); rew_T = torch.empty((STEPS, n), dtype=torch.float64, device=dev)
    done_T = torch.empty((STEPS, n), dtype=torch.uint8, device=dev)
    worst = 0.0; ndone = 0; max_nefc = 0; wrapped = np.zeros(n, dtype=bool)
    if horizon:
        b.rollout(acts, (obs_T, rew_T, done_T), 1)
        b.join(); torch.cuda.current_stream().synchronize()
    if queue:
        for t in range(STEPS):
            b.step(acts[t], 1, (obs_T[t], rew_T[t], done_T[t]))
        assert b.queue_stats() == (0, 0, STEPS)
        b.join(); torch.cuda.current_stream().synchronize()
        assert b.queue_stats() == (1, STEPS, 0)
        horizon = True                                            # from here on: the batch's fields are those of the last step
    for t in range(STEPS):
        if not horizon:
            b.step(acts[t], 1, (obs_T[t], rew_T[t], done_T[t]))  # pipelined: two sub-batch launches on their own streams
            b.join(); torch.cuda.current_stream().synchronize()
        a = acts[t].cpu().numpy(); obs = obs_T[t].cpu().numpy(); rew = rew_T[t].cpu().numpy(); done = done_T[t].cpu().numpy()
        o_obs, o_rew, o_done = O.batch_step_imitation(om, ods, a, 1, T, P, fidx, cyc, nthreads=nthreads)
        assert np.array_equal(done, o_done), "done flags differ at step %d: envs %s" % (t, np.nonzero(done != o_done)[0][:8])
        state_now = (not horizon) or t == STEPS - 1              # the batch's fields are those of step t (horizon launch: of the last step only)
        # sim.data.* as they stand after sim.step(): row / contact counts and the contact list of the 4th RK stage, every env
        o_nefc = np.array([int(d.get("nefc")[0]) for d in ods], dtype=np.int32)
        if state_now:
            nefc = b.get(A.F_NEFC); ncon = b.get(A.F_NCON); cg = b.get(A.F_CONTACT_GEOMS)
            o_ncon = np.array([int(d.get("ncon")[0]) for d in ods], dtype=np.int32)
            assert np.array_equal(nefc, o_nefc), "nefc differs at step %d: envs %s" % (t, np.nonzero(nefc != o_nefc)[0][:8])
            assert np.array_equal(ncon, o_ncon), "ncon differs at step %d" % t
            for e in range(n):
                k = min(int(o_ncon[e]), A.MAXEFC)
                if k:
                    ocg = ods[e].get("contact_geom").reshape(-1, 2).astype(np.int32)
                    assert np.array_equal(cg[e][:k], ocg[:k]), "contact (geom1, geom2) list differs: step %d env %d" % (t, e)
                assert np.all(cg[e][k:] == -1)
        max_nefc = max(max_nefc, int(o_nefc.max()))
        wrapped |= cyc > 0                                        # the clip's last frame was passed inside an episode (cycle counter, root shift of the reference pose)
        # mirror of the device's auto-reset: hard reset onto the frame its RNG draws for (seed, global id, episode)
        dn = np.nonzero(done)[0]
        for e in dn:
            k = H.device_rsi_frame(SEED, off + int(e), int(episode[e]), F)
            episode[e] += 1
            ods[e].reset(); ods[e].set_state(mc.data_config[k], mc.data_vel[k])
            fidx[e] = k; cyc[e] = 0
            o_obs[e] = np.concatenate([mc.data_config[k][7:], mc.data_vel[k][6:]])      # DummyVecEnv convention: the fresh episode's obs
        ndone += len(dn)
        err_o = np.abs(obs - o_obs).max(1) / np.maximum(1.0, np.abs(o_obs).max(1))
        err_r = np.abs(rew - o_rew) / np.maximum(1.0, np.abs(o_rew))
        worst = max(worst, float(err_o.max()), float(err_r.max()))
        assert err_o.max() < 1e-9, "obs differ at step %d: env %d rel err %.3e" % (t, int(err_o.argmax()), err_o.max())
        assert err_r.max() < 1e-9, "reward differs at step %d: env %d" % (t, int(err_r.argmax()))
        if state_now:
            assert np.array_equal(b.get(A.F_FRAME_IDX), fidx), "frame cursors differ at step %d" % t
            assert np.array_equal(b.get(A.F_CYCLE), cyc), "cycle counters differ at step %d" % t
            assert np.array_equal(b.get(A.F_EPISODE), episode.astype(np.int32))
    q = b.get(A.F_QPOS); w = b.get(A.F_QACC_WARMSTART); tm = b.get(A.F_TIME)
    oq = np.stack([d.get("qpos") for d in ods]); ow = np.stack([d.get("qacc_warmstart") for d in ods]); ot = np.array([d.get("time")[0] for d in ods])
    assert np.abs(q - oq).max() / max(1.0, np.abs(oq).max()) < 1e-9
    werr = float(np.abs(w - ow).max() / max(1.0, np.abs(ow).max()))
    assert werr < 1e-8, "qacc_warmstart rel err %.3e" % werr           # accelerations: conditioned like the contact solve (round 6: 1e-7 -> 1e-8)
    assert np.abs(tm - ot).max() < 1e-12
    assert ndone > 0 and max_nefc > 16, "the run must contain early terminations and heavy contact (%d done, max nefc %d)" % (ndone, max_nefc)
    if STEPS >= 64:
        assert int((episode >= 2).sum()) > n // 2, "a 64-step run must reach second-episode RSI draws for most environments"
    if STEPS == 256:
        assert int((episode >= 4).sum()) > n // 2, "a 256-step run holds several episodes per environment"
    if STEPS >= 64:
        assert int(wrapped.sum()) > 0, "a %d-step run of a %d-frame clip with RSI starts must take environments past the clip's last frame" % (STEPS, F)
    assert (b.get(A.F_STATUS) & 1).sum() == 0
    if packed:
        print("   packed: env-steps handed to the one-env code [total, candidates, box slots, contacts, rows, PGS test]:", b.redo_reasons())
    print("full shard %s x %d, %d steps: worst rel err %.2e (qacc_warmstart %.2e), %d auto-resets, %d envs wrapped the clip (%d frames), max nefc %d, oracle threads %d"
          % (clip, n, STEPS, worst, werr, ndone, int(wrapped.sum()), F, max_nefc, nthreads))
    b.close()


@pytest.mark.parametrize("form", ["packed", "horizon"])
def test_config2_full_size_p_controller_matches_oracle_every_env(form):
    """BASELINE.json configs[1] as bench.py --workload cfg2 runs it — 'walk', 4 096 envs, contacts and joint limits off, the P-controller of
    src/env_torque_test.py:14-20 as the action front-end (ctrl = 0.8 (mocap_cfg[idx][7:] - qpos[7:]) + action), RSI auto-reset when the
    falling body leaves the height band — on the kernel that is timed for it (k_step_packed, four environments per wavefront, two pipelined
    sub-batches) and through one horizon launch, against the oracle: EVERY env, 16 steps, obs 1e-9, done flags, the unclamped ctrl."""
    import torch
    from deepmimic_mujoco_amd import Batch
    from oracle import oracle as O
    n, STEPS, off = 4096, 16, 0
    mc = H.mocap("walk")
    F = mc.data_config.shape[0]
    b = Batch(H.compiled_model(), mc.data_config, mc.data_vel, n, device=0, flags=A.FLAG_NO_CONTACT | A.FLAG_NO_LIMIT, mocap_dt=float(mc.dt))
    b.set_option(A.OPT_REWARD_MODE, 0); b.set_option(A.OPT_AUTORESET, 1); b.set_option(A.OPT_SEED, SEED); b.set_option(A.OPT_ACTION_MODE, 1)
    b.set_option(A.OPT_PIPELINE, 2); b.set_option(A.OPT_PACKED, 1)
    if form == "horizon":
        b.set_option(106, 1)                                     # (a rowless model steps per launch by default: every wave costs the same)
    b.reset(0, 1)
    fidx = b.get(A.F_FRAME_IDX).copy()
    assert np.array_equal(fidx, np.array([H.device_rsi_frame(SEED, off + e, 0, F) for e in range(n)], dtype=np.int32))
    om = H.oracle_model(enable_contact=0, enable_limit=0)
    ods = [O.Data(om) for _ in range(n)]
    for e in range(n):
        ods[e].reset(); ods[e].set_state(mc.data_config[fidx[e]], mc.data_vel[fidx[e]])
    episode = np.ones(n, dtype=np.int64)
    nthreads = max(1, len(os.sched_getaffinity(0)))
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cuda"); g.manual_seed(6)
    acts = torch.randn((STEPS + 1, n, 28), generator=g, device=dev, dtype=torch.float64) * 0.2
    obs_T = torch.empty((STEPS, n, 56), dtype=torch.float64, device=dev); rew_T = torch.empty((STEPS, n), dtype=torch.float64, device=dev)
    done_T = torch.empty((STEPS, n), dtype=torch.uint8, device=dev)
    if form == "horizon":
        b.rollout(acts, (obs_T, rew_T, done_T), 1)
    else:
        for t in range(STEPS):
            b.step(acts[t], 1, (obs_T[t], rew_T[t], done_T[t]))
    b.join(); torch.cuda.current_stream().synchronize()
    ndone = 0; worst = 0.0
    for t in range(STEPS):
        a = acts[t].cpu().numpy(); obs = obs_T[t].cpu().numpy(); rew = rew_T[t].cpu().numpy(); done = done_T[t].cpu().numpy()
        qnow = np.stack([d.get("qpos") for d in ods])
        ctrl = a + 0.8 * (mc.data_config[fidx][:, 7:] - qnow[:, 7:])
        o_obs, o_rew, o_done = O.batch_step(om, ods, ctrl, 1, nthreads=nthreads)
        assert np.array_equal(done, o_done), "done flags differ at step %d" % t
        for e in np.nonzero(done)[0]:
            k = H.device_rsi_frame(SEED, off + int(e), int(episode[e]), F)
            episode[e] += 1
            ods[e].reset(); ods[e].set_state(mc.data_config[k], mc.data_vel[k])
            fidx[e] = k
            o_obs[e] = np.concatenate([mc.data_config[k][7:], mc.data_vel[k][6:]])
        ndone += int(done.sum())
        err = np.abs(obs - o_obs).max(1) / np.maximum(1.0, np.abs(o_obs).max(1))
        worst = max(worst, float(err.max()))
        assert err.max() < 1e-9, "obs differ at step %d: env %d rel err %.3e" % (t, int(err.argmax()), err.max())
        assert np.array_equal(rew, o_rew)
        if t == STEPS - 1:
            assert np.abs(b.get(A.F_CTRL) - ctrl).max() < 1e-9       # data.ctrl keeps the unclamped controller output
    assert np.array_equal(b.get(A.F_FRAME_IDX), fidx) and np.array_equal(b.get(A.F_EPISODE), episode.astype(np.int32))
    assert np.all(b.get(A.F_NEFC) == 0) and b.redo_total() == 0
    q = b.get(A.F_QPOS); oq = np.stack([d.get("qpos") for d in ods])
    assert np.abs(q - oq).max() / max(1.0, np.abs(oq).max()) < 1e-9
    print("configs[1] full size (%s): worst rel err %.2e, %d auto-resets" % (form, worst, ndone))
    b.close()


@pytest.mark.parametrize("form", ["packed", "packed-ext", "horizon", "one-env"])
def test_standing_population_full_shard_matches_oracle(form):
    """The regime of a competent policy (src/checkpoint_tmp/DeepMimic/trpo-walk-0: under it 23 % of evaluations hold more than 16 rows):
    4 096 environments started from the noisy init pose (src/dp_env_v3.py:158-164), standing on both feet — 8 foot corners x 4 pyramid edges =
    32 constraint rows, plus whatever joint limits are active — and kept there for 16 steps by small actions.  Most packed waves take the
    two-row-set path, environments beyond 32 rows go through the redo list (per step), stay in their wave (per step with the three-set code,
    OPT_PACKED 2: "packed-ext"; inside a horizon launch where the wave predicted them) or take the in-wave re-step (horizon launch, mispredicted).
    Every env, every step against the oracle: obs / reward 1e-9, done flags, row and contact counts, contact lists."""
    import torch
    from deepmimic_mujoco_amd import Batch
    from deepmimic_mujoco_amd.imitation import ImitationSpec
    from oracle import oracle as O
    n, STEPS, clip = 4096, 16, "walk"
    mc = H.mocap(clip)
    T, P = ImitationSpec(H.compiled_model()).table_for(mc)
    b = Batch(H.compiled_model(), mc.data_config, mc.data_vel, n, device=0, mocap_dt=float(mc.dt), imitation=(T, P))
    b.set_option(A.OPT_REWARD_MODE, 3); b.set_option(A.OPT_AUTORESET, 0); b.set_option(A.OPT_SEED, SEED); b.set_option(A.OPT_DIAGNOSTICS, 1)
    b.set_option(A.OPT_PIPELINE, 2); b.set_option(A.OPT_PACKED, 0 if form == "one-env" else 2 if form == "packed-ext" else 1)
    b.reset(1, 1)                                                 # reset_model_init after sim.reset(): init pose + U(-0.01, 0.01) noise, frame redrawn
    q0 = b.get(A.F_QPOS); v0 = b.get(A.F_QVEL)
    fidx = b.get(A.F_FRAME_IDX).copy(); cyc = b.get(A.F_CYCLE).copy()
    assert np.abs(q0[:, 2] - 0.9).max() < 0.011 and np.abs(v0).max() < 0.011
    om = H.oracle_model()
    ods = [O.Data(om) for _ in range(n)]
    for e in range(n):
        ods[e].reset(); ods[e].set_state(q0[e], v0[e])
    nthreads = max(1, len(os.sched_getaffinity(0)))
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cuda"); g.manual_seed(8)
    acts = torch.randn((STEPS + 1, n, 28), generator=g, device=dev, dtype=torch.float64) * 0.1
    obs_T = torch.empty((STEPS, n, 56), dtype=torch.float64, device=dev); rew_T = torch.empty((STEPS, n), dtype=torch.float64, device=dev)
    done_T = torch.empty((STEPS, n), dtype=torch.uint8, device=dev)
    if form == "horizon":
        b.rollout(acts, (obs_T, rew_T, done_T), 1)
        b.join(); torch.cuda.current_stream().synchronize()
    worst = 0.0
    peak = np.zeros(n, dtype=np.int32)
    above32 = near_cap = 0                   # env-steps whose last evaluation held more than 32 rows / came within two rows of the packed capacity
    for t in range(STEPS):
        if form != "horizon":
            b.step(acts[t], 1, (obs_T[t], rew_T[t], done_T[t]))
            b.join(); torch.cuda.current_stream().synchronize()
        a = acts[t].cpu().numpy(); obs = obs_T[t].cpu().numpy(); rew = rew_T[t].cpu().numpy(); done = done_T[t].cpu().numpy()
        o_obs, o_rew, o_done = O.batch_step_imitation(om, ods, a, 1, T, P, fidx, cyc, nthreads=nthreads)
        assert np.array_equal(done, o_done), "done flags differ at step %d" % t
        o_nefc = np.array([int(d.get("nefc")[0]) for d in ods], dtype=np.int32)
        peak = np.maximum(peak, o_nefc)
        above32 += int((o_nefc > 32).sum()); near_cap += int((o_nefc >= A.PACKED_MAXROWS - 2).sum())
        if form != "horizon" or t == STEPS - 1:
            nefc = b.get(A.F_NEFC); ncon = b.get(A.F_NCON); cg = b.get(A.F_CONTACT_GEOMS)
            o_ncon = np.array([int(d.get("ncon")[0]) for d in ods], dtype=np.int32)
            assert np.array_equal(nefc, o_nefc), "nefc differs at step %d: envs %s" % (t, np.nonzero(nefc != o_nefc)[0][:8])
            assert np.array_equal(ncon, o_ncon)
            for e in range(n):
                k = min(int(o_ncon[e]), A.MAXEFC)
                if k:
                    assert np.array_equal(cg[e][:k], ods[e].get("contact_geom").reshape(-1, 2).astype(np.int32)[:k]), "contact list differs: step %d env %d" % (t, e)
        err_o = np.abs(obs - o_obs).max(1) / np.maximum(1.0, np.abs(o_obs).max(1))
        err_r = np.abs(rew - o_rew) / np.maximum(1.0, np.abs(o_rew))
        worst = max(worst, float(err_o.max()), float(err_r.max()))
        assert err_o.max() < 1e-9, "obs differ at step %d: env %d rel err %.3e (nefc %d)" % (t, int(err_o.argmax()), err_o.max(), int(o_nefc[int(err_o.argmax())]))
        assert err_r.max() < 1e-9, "reward differs at step %d" % t
    heavy = float((peak >= 32).mean())
    assert heavy >= 0.25, "a standing population: %.1f %% of the environments reached 32 rows" % (100 * heavy)
    assert int(done_T.sum()) == 0
    q = b.get(A.F_QPOS); oq = np.stack([d.get("qpos") for d in ods])
    assert np.abs(q - oq).max() / max(1.0, np.abs(oq).max()) < 1e-9
    print("standing shard (%s): worst rel err %.2e; %.1f %% of envs reached >= 32 rows, %.1f %% > 32 (max %d); beyond the packed capacities: %s"
          % (form, worst, 100 * heavy, 100 * float((peak > 32).mean()), int(peak.max()), b.redo_reasons() if form != "one-env" else "-"))
    if form == "packed-ext":
        rr = b.redo_reasons()
        assert rr[4] <= near_cap and rr[0] < 0.01 * above32, "the three-set per-step kernel keeps every env-step within 40 rows in its wave: %s of %d" % (rr, above32)
    if form == "horizon":
        # round 5: inside a horizon launch 33 .. 40 rows (both feet flat + joint limits) are solved by the packed path itself (slot_kernel.h
        # slot_constraint<3>, the step's second instantiation): the population's environments above 32 rows were NOT handed to the one-env code
        rr = b.redo_reasons()
        print("   env-steps above 32 rows: %d; within two rows of the capacity (%d): %d; handed to the one-env code for rows: %d" % (above32, A.PACKED_MAXROWS, near_cap, rr[4]))
        assert above32 > 0.01 * n * STEPS, "the standing population must hold environments above 32 rows"
        # (not all: a wave calls the three-set instantiation of the step while one of its environments held 24+ rows after the LAST step — the step
        #  that takes an environment from fewer rows past 32, e.g. the landing from the init pose this population starts with, is re-stepped as before)
        assert rr[0] < 0.1 * above32, "nine in ten env-steps above 32 rows must stay on the packed path: %s of %d" % (rr, above32)
    b.close()
