"""Policy / rollout / GAE host logic and the statistical anchor of the physics restatement:
the reference's shipped TRPO policy (trained inside real MuJoCo) replayed in the oracle reproduces the episode lengths its
own training log reports."""
import os
import shutil

import numpy as np
import pytest
import torch

from deepmimic_mujoco_amd import _abi as A
from deepmimic_mujoco_amd.policy import MlpPolicy, RunningMeanStd
from deepmimic_mujoco_amd.rollout import add_vtarg_and_adv, traj_segment_generator, flatten_segment
from deepmimic_mujoco_amd.tf_checkpoint import load_checkpoint, read_index, crc32c
from tests import helpers as H

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CKPT = os.path.join(GOLD, "ckpt", "trpo-walk-0")


def test_crc32c_known_answers():
    assert crc32c(b"") == 0
    assert crc32c(b"123456789") == 0xE3069283          # the standard CRC-32C check value
    assert crc32c(bytes(32)) == 0x8A9136AA             # RFC 3720 B.4: 32 zero bytes


def test_tf_checkpoint_reader_shipped_bundle(tmp_path):
    idx = read_index(CKPT)
    assert len(idx) == 32 and sorted(k.split("/")[0] for k in idx)[0] == "oldpi"
    assert idx["pi/polfc1/w"][:2] == (np.dtype("<f4"), (56, 100))
    assert idx["pi/obfilter/count"][:2] == (np.dtype("<f8"), ())
    d = load_checkpoint(CKPT, scope="pi")                # verifies every tensor's crc32c
    assert set(d) == {"logstd", "obfilter/count", "obfilter/runningsum", "obfilter/runningsumsq"} | {
        "%s/%s" % (l, p) for l in ("polfc1", "polfc2", "polfinal", "vffc1", "vffc2", "vffinal") for p in ("w", "b")}
    assert d["polfinal/w"].shape == (100, 28) and d["vffinal/w"].shape == (100, 1) and d["logstd"].shape == (1, 28)
    # a flipped data byte is caught by the checksum
    for ext in (".index", ".data-00000-of-00001"):
        shutil.copyfile(CKPT + ext, str(tmp_path / ("c" + ext)))
    p = str(tmp_path / "c.data-00000-of-00001")
    raw = bytearray(open(p, "rb").read()); raw[200000] ^= 0x40
    open(p, "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="crc32c"):
        load_checkpoint(str(tmp_path / "c"))


def test_policy_forward_matches_numpy_restatement():
    g = np.load(os.path.join(GOLD, "policy_forward_golden.npz"))
    pol = MlpPolicy.from_tf_checkpoint(CKPT)
    mean, vpred = pol.forward(torch.from_numpy(g["ob"]))
    assert np.abs(mean.numpy() - g["mean"]).max() < 2e-5 * max(1.0, np.abs(g["mean"]).max())
    assert np.abs(vpred.numpy() - g["vpred"]).max() < 2e-5 * max(1.0, np.abs(g["vpred"]).max())
    ac, vp = pol.act(False, torch.from_numpy(g["ob"]))
    assert ac.dtype == torch.float64 and torch.equal(ac, mean.to(torch.float64)) and torch.equal(vp, vpred)
    a1, v1 = pol.act(False, torch.from_numpy(g["ob"][3]))            # single observation like the reference's act()
    assert a1.shape == (28,) and abs(float(v1) - float(vpred[3])) < 1e-5


def test_policy_entropy_matches_training_log():
    log = np.load(os.path.join(GOLD, "trpo_walk0_log.npz"))
    pol = MlpPolicy.from_tf_checkpoint(CKPT)
    assert abs(pol.entropy() - log["entropy"][-1]) < 0.05           # 35.72 in the checkpoint vs 35.73 logged at the last iteration


def test_policy_sampling_and_neglogp():
    pol = MlpPolicy.from_tf_checkpoint(CKPT)
    pol.seed(3)
    ob = torch.zeros((20000, 56), dtype=torch.float64)
    ac, _ = pol.act(True, ob)
    mean, _ = pol.forward(ob)
    resid = (ac.to(torch.float32) - mean) / torch.exp(pol.params["logstd"])
    assert abs(float(resid.mean())) < 0.01 and abs(float(resid.std()) - 1.0) < 0.01
    nl = pol.neglogp(ob[:5], mean[:5])                                # at the mode: 0.5*k*log(2 pi) + sum(logstd)
    assert torch.allclose(nl, torch.full((5,), 0.5 * np.log(2 * np.pi) * 28 + float(pol.params["logstd"].sum())), atol=1e-4)
    fresh = MlpPolicy(seed=1)                                          # reference initialisation
    for name, s in (("polfc1/w", 1.0), ("vffc2/w", 1.0), ("polfinal/w", 0.01)):
        assert torch.allclose(torch.sqrt((fresh.params[name] ** 2).sum(0)), torch.full((fresh.params[name].shape[1],), s), atol=1e-5)
    assert float(fresh.params["logstd"].abs().max()) == 0.0
    rt = MlpPolicy().load_state_dict(pol.state_dict())
    assert torch.equal(rt.forward(ob[:3])[0], pol.forward(ob[:3])[0])


def test_running_mean_std():
    rms = RunningMeanStd((5,))
    rng = np.random.RandomState(0)
    xs = [rng.randn(17, 5) * 3 + 1, rng.randn(40, 5) * 0.01]
    for x in xs:
        rms.update(torch.from_numpy(x))
    allx = np.concatenate(xs)
    cnt = 1e-2 + len(allx)
    mean = (allx.sum(0) / cnt).astype(np.float32)
    std = np.sqrt(np.maximum(((allx ** 2).sum(0) + 1e-2) / cnt - mean.astype(np.float64) ** 2, 1e-2))
    assert np.allclose(rms.mean.numpy(), mean, atol=1e-6) and np.allclose(rms.std.numpy(), std, atol=1e-5)
    assert float(RunningMeanStd((3,)).std[0]) == pytest.approx(1.0)   # sqrt(max(eps/eps - 0, 1e-2))


def _gae_reference_loop(rew, vpred, new, nextvpred, gamma, lam):
    """numpy restatement of src/trpo.py:83-94 for ONE environment."""
    new = np.append(new, 0)
    vpred = np.append(vpred, nextvpred)
    T = len(rew)
    gaelam = np.empty(T, "float32")
    last = 0
    for t in reversed(range(T)):
        nonterminal = 1 - new[t + 1]
        delta = rew[t] + gamma * vpred[t + 1] * nonterminal - vpred[t]
        gaelam[t] = last = delta + gamma * lam * nonterminal * last
    return gaelam, gaelam + vpred[:-1]


def test_gae_matches_reference_loop():
    rng = np.random.RandomState(1)
    T, N = 64, 9
    rew = rng.rand(T, N).astype(np.float32)
    vpred = rng.randn(T, N).astype(np.float32) * 5
    new = (rng.rand(T, N) < 0.1).astype(np.int32); new[0] = 1
    nxt = rng.randn(N).astype(np.float32)
    seg = {"rew": torch.from_numpy(rew), "vpred": torch.from_numpy(vpred), "new": torch.from_numpy(new), "nextvpred": torch.from_numpy(nxt)}
    add_vtarg_and_adv(seg, 0.995, 0.97)
    for e in range(N):
        adv, ret = _gae_reference_loop(rew[:, e], vpred[:, e], new[:, e], nxt[e], 0.995, 0.97)
        assert np.allclose(seg["adv"][:, e].numpy(), adv, rtol=2e-5, atol=2e-5)
        assert np.allclose(seg["tdlamret"][:, e].numpy(), ret, rtol=2e-5, atol=2e-5)
    flat = flatten_segment(seg)
    assert flat["adv"].shape == (T * N,) and torch.equal(flat["adv"][T:2 * T], seg["adv"][:, 1])


def _emu_vec_env(n, seed):
    from deepmimic_mujoco_amd import DPVecEnv
    from tests.emu.emu import EmuBatch
    return DPVecEnv(n, motion="walk", reward="alive", autoreset="init", seed=seed,
                    batch_factory=lambda cm, cfg, vel, ne, flags: EmuBatch(cm, cfg, vel, ne, flags))


def test_segment_generator_equals_per_step_loop_on_testbench():
    """The batched generator (kernel source on the wave testbench) against the reference's loop written out by hand."""
    n, T = 3, 4
    pol = MlpPolicy.from_tf_checkpoint(CKPT)
    env = _emu_vec_env(n, 5)
    pol.seed(11)
    gen = traj_segment_generator(pol, env, T, stochastic=True)
    seg1 = {k: (v.clone() if torch.is_tensor(v) else list(v)) for k, v in next(gen).items()}
    seg2 = next(gen)

    env2 = _emu_vec_env(n, 5)
    pol.seed(11)
    ob = torch.from_numpy(env2.reset("rsi"))
    new = np.ones(n, np.int32)
    rows = []
    for t in range(2 * T + 1):
        ac, vpred = pol.act(True, ob)
        rows.append((ob.clone(), ac.clone(), vpred.clone(), new.copy()))
        o, r, d = env2.batch.step(ac.numpy(), 1)
        rows[-1] += (r.copy(),)
        ob = torch.from_numpy(o); new = d.astype(np.int32)
    for k, seg in enumerate((seg1, seg2)):
        for i in range(T):
            ob_i, ac_i, vp_i, new_i, r_i = rows[k * T + i]
            assert torch.equal(seg["ob"][i], ob_i.to(torch.float32))
            assert torch.equal(seg["ac"][i], ac_i.to(torch.float32))
            assert torch.equal(seg["vpred"][i], vp_i)
            assert np.array_equal(seg["new"][i].numpy(), new_i)
            assert np.array_equal(seg["rew"][i].numpy(), r_i.astype(np.float32))
        nxt = rows[(k + 1) * T]
        assert torch.allclose(seg["nextvpred"], nxt[2] * torch.from_numpy(1 - nxt[3]).to(torch.float32))
    assert torch.equal(seg2["prevac"][1], seg2["ac"][0]) and torch.equal(seg2["prevac"][0], seg1["ac"][T - 1])
    assert seg1["new"][0].tolist() == [1] * n


def _first_episode_lengths(pol, stochastic, n, cap, seed):
    """Protocol of src/trpo.py:27-80 on the oracle: noisy default pose (reset_model_init), one env.step per policy action,
    until is_done.  Returns the first-episode length per env (cap = censored)."""
    from oracle import oracle as O
    om = O.Model()
    ds = [O.Data(om) for _ in range(n)]
    rng = np.random.RandomState(seed)
    for d in ds:
        d.reset()
        d.set_state(d.get("qpos") + rng.uniform(-0.01, 0.01, 35), d.get("qvel") + rng.uniform(-0.01, 0.01, 34))
    length = np.zeros(n, int)
    alive = np.ones(n, bool)
    pol.seed(seed)
    for t in range(cap):
        ob = np.stack([np.concatenate([d.get("qpos")[7:], d.get("qvel")[6:]]) for d in ds])
        ac, _ = pol.act(stochastic, torch.from_numpy(ob))
        _o, _r, done = O.batch_step(om, ds, ac.numpy(), 1, min(n, os.cpu_count() or 1))
        fin = alive & (done != 0)
        length[fin] = t + 1
        alive &= ~fin
        if not alive.any():
            break
    length[alive] = cap
    return length


def test_shipped_policy_reproduces_training_log_episode_lengths():
    """Statistical anchor: the run that produced the shipped checkpoint logged EpLenMean ~ 33-37 for
    the untrained policy and ~ 235-287 over its last 100 iterations (real MuJoCo 2.0).  The same policies in the oracle's
    physics: 34 and ~ 290.  A wrong contact / limit / actuator model cannot balance a policy trained against the real one."""
    log = np.load(os.path.join(GOLD, "trpo_walk0_log.npz"))["EpLenMean"]
    assert 30 < log[:5].mean() < 40 and 230 < log[-100:].mean() < 290
    untrained = _first_episode_lengths(MlpPolicy(seed=0), True, 32, 200, 0)
    assert abs(untrained.mean() - log[:5].mean()) < 6, untrained.mean()
    trained = _first_episode_lengths(MlpPolicy.from_tf_checkpoint(CKPT), True, 48, 1500, 0)
    assert 0.7 * log[-100:].mean() < trained.mean() < 1.5 * log[-100:].mean(), trained.mean()


class _ScriptedEnv:
    """Host stand-in for DPVecEnv with scripted rewards / dones: checks the generator's bookkeeping only."""

    def __init__(self, n, steps, seed):
        rng = np.random.RandomState(seed)
        self.num_envs = n
        self.rew = rng.rand(steps, n)
        self.done = (rng.rand(steps, n) < 0.15).astype(np.uint8)
        self.t = 0
        self.batch = self

    def reset(self, mode, out=None):
        out[...] = 0.0
        return out

    def step(self, ac, nsub, out):
        ob, rew, done = out
        ob[...] = self.t + 1
        rew[...] = self.rew[self.t]; done[...] = self.done[self.t]
        self.t += 1
        return out


def test_segment_episode_statistics_across_segment_boundaries():
    n, T, segs = 7, 16, 4
    env = _ScriptedEnv(n, T * segs, 3)
    gen = traj_segment_generator(MlpPolicy(), env, T, stochastic=False)
    got_rets, got_lens, news = [], [], []
    for _ in range(segs):
        seg = next(gen)
        got_rets += seg["ep_rets"]; got_lens += seg["ep_lens"]; news.append(seg["new"].numpy().copy())
        assert float(seg["ob"][3, 0, 0]) == float(len(news) - 1) * T + 3          # row t holds the observation seen at step t
    # the reference's single-env bookkeeping (src/trpo.py:70-76), replayed env by env in time-major order
    want = []
    cur_ret = np.zeros(n); cur_len = np.zeros(n, int)
    for t in range(T * segs):
        cur_ret += env.rew[t]; cur_len += 1
        for e in range(n):
            if env.done[t, e]:
                want.append((cur_ret[e], cur_len[e])); cur_ret[e] = 0; cur_len[e] = 0
    assert got_lens == [w[1] for w in want]
    assert np.allclose(got_rets, [w[0] for w in want], rtol=1e-12, atol=1e-12)
    new = np.concatenate(news)
    assert np.array_equal(new[0], np.ones(n, int)) and np.array_equal(new[1:], env.done[:T * segs - 1])


def test_reference_log_formats_round_trip(tmp_path):
    """progress.csv / monitor.csv: parse the head of the reference's own files (fixtures are verbatim
    excerpts of src/log_tmp/DeepMimic/trpo-walk-0/*), write the same content with our writers, get identical bytes back."""
    from deepmimic_mujoco_amd.logio import ProgressCsv, MonitorWriter, read_progress_csv, read_monitor_csv
    ref = os.path.join(GOLD, "ref_progress_head.csv")
    kv = read_progress_csv(ref)
    assert list(kv)[0] == "EpRewMean" and kv["EpLenMean"][0] == 36.8 and len(kv["meankl"]) == 40
    raw = open(ref).read().splitlines()
    w = ProgressCsv(str(tmp_path / "progress.csv"))
    keys = raw[0].split(",")
    for line in raw[1:]:
        w.writekvs(dict(zip(keys, line.split(","))))          # cells as the reference printed them
    w.close()
    assert open(str(tmp_path / "progress.csv")).read().splitlines() == raw
    w = ProgressCsv(str(tmp_path / "p2.csv")); w.writekvs({"a": 1}); w.writekvs({"a": 2, "b": 3}); w.close()
    assert open(str(tmp_path / "p2.csv")).read() == "a,b\n1,\n2,3\n"       # late keys pad earlier rows (logger.py:110-124)
    hdr, r, l, t = read_monitor_csv(os.path.join(GOLD, "ref_monitor_head.csv"))
    assert hdr["env_id"] is None and r[:3] == [18.0, 31.0, 25.0] and l[:3] == [18, 31, 25] and len(l) == 200
    m = MonitorWriter(str(tmp_path / "run"), t_start=hdr["t_start"])
    for ri, li, ti in zip(r, l, t):
        m.write_episodes([ri], [li], t=hdr["t_start"] + ti)
    m.close()
    h2, r2, l2, t2 = read_monitor_csv(str(tmp_path / "run.monitor.csv"))
    assert h2 == hdr and r2 == r and l2 == l and np.allclose(t2, t, atol=1e-6)


def test_tf_checkpoint_writer_reproduces_the_shipped_bundle_byte_for_byte(tmp_path):
    """`save_checkpoint` is the inverse of the reader on the reference's own artefact: re-saving the 32 tensors of
    src/checkpoint_tmp/DeepMimic/trpo-walk-0 gives identical `.index` (table blocks, prefix compression, restart points, masked
    crc32c, footer) and `.data` files, so a bundle written here is one `tf.train.Saver.restore` reads."""
    from deepmimic_mujoco_amd.tf_checkpoint import save_checkpoint
    d = load_checkpoint(CKPT)
    out = str(tmp_path / "resaved")
    save_checkpoint(out, d)
    for ext in (".index", ".data-00000-of-00001"):
        assert open(out + ext, "rb").read() == open(CKPT + ext, "rb").read(), ext
    assert 'model_checkpoint_path: "resaved"' in open(str(tmp_path / "checkpoint")).read()


def test_policy_saved_as_tf_bundle_round_trips_and_has_the_reference_layout(tmp_path):
    pol = MlpPolicy(seed=4)
    pol.ob_rms.update(torch.randn(300, 56, dtype=torch.float64) * 2 + 0.3)
    with torch.no_grad():
        pol.params["logstd"] -= 0.25
    pre = str(tmp_path / "DeepMimic" / "trpo-walk-7")
    pol.save_tf_checkpoint(pre)
    idx = read_index(pre); ref = read_index(CKPT)
    assert sorted(idx) == sorted(ref)                                           # same 32 variables ...
    for k in ref:                                                                # ... same dtype, shape, offset and size
        assert idx[k][:5] == ref[k][:5], k
    back = MlpPolicy.from_tf_checkpoint(pre)
    for k, v in pol.params.items():
        assert torch.equal(back.params[k], v.detach()), k
    assert torch.equal(back.ob_rms.sum, pol.ob_rms.sum) and torch.equal(back.ob_rms.count, pol.ob_rms.count)
    ob = torch.randn(5, 56, dtype=torch.float64)
    assert torch.equal(back.forward(ob)[0], pol.forward(ob)[0])
    old = load_checkpoint(pre, scope="oldpi")
    assert np.array_equal(old["polfc1/w"], pol.params["polfc1/w"].detach().numpy())
    # a bigger bundle spills into several table blocks: still read back exactly (multi-block index, separators)
    from deepmimic_mujoco_amd.tf_checkpoint import save_checkpoint
    big = {"scope%03d/some/rather/long/variable/name_%d" % (i, i): np.full((i % 5 + 1, 3), i, dtype=np.float32) for i in range(300)}
    big["z/i64"] = np.arange(7, dtype=np.int64); big["a/flag"] = np.array([True, False]); big["a/scalar"] = np.float64(3.5)
    save_checkpoint(str(tmp_path / "big"), big)
    got = load_checkpoint(str(tmp_path / "big"))
    assert sorted(got) == sorted(big) and all(np.array_equal(got[k], np.asarray(big[k])) and got[k].dtype == np.asarray(big[k]).dtype for k in big)


def test_fused_rollout_step_is_refused_without_the_native_path():
    """`SegmentCollector(fused=True)` needs the HIP policy epilogue: on CPU tensors / without a device-resident env batch it raises
    instead of silently running something else; `can_fuse` says so beforehand."""
    from deepmimic_mujoco_amd.rollout import SegmentCollector, can_fuse

    class Env:
        num_envs = 2
        batch = object()

        def reset(self, mode, out=None):
            return out

    pi = MlpPolicy(seed=0)
    assert not can_fuse(pi, Env())
    with pytest.raises(ValueError):
        SegmentCollector(pi, Env(), 4, fused=True)


def test_collector_kernel_choice_is_a_function_of_the_last_horizons_statistics():
    """`SegmentCollector._choose_kernel` (host logic, no device): starts packed, hands a batch to the one-env steps when more than
    HORIZON_REDO_RATE_MAX of the last horizon's env-steps overflowed the packed path, hands it back when no environment holds more than
    HORIZON_HEAVY_ROWS constraint rows, and leaves batches alone whose configuration pins the kernel."""
    import numpy as np
    from deepmimic_mujoco_amd import _abi as A
    from deepmimic_mujoco_amd.rollout import SegmentCollector

    class FakeBatch:
        REDO_RATE_MAX, HEAVY_ROWS = 3e-4, 30

        def __init__(self):
            self.options, self.redo, self.nefc, self._auto = {}, 0, np.zeros(8, dtype=np.int32), True

        def set_option(self, o, v):
            self.options[int(o)] = int(v)

        def enable_auto_packed(self, on):
            self._auto = bool(on)

        def rebaseline_auto(self):
            pass

        def redo_total(self):
            return self.redo

        def get(self, f):
            assert f == A.F_NEFC
            return self.nefc

    class FakeEnv:
        def __init__(self, ok):
            self.batch, self.horizon_packed_ok = FakeBatch(), ok

    c = SegmentCollector.__new__(SegmentCollector)
    c.env, c.T, c.n, c._redo_seen, c.kernel_switches = FakeEnv(True), 100, 8, None, 0
    b = c.env.batch
    c._choose_kernel()                                                   # first horizon: packed, the per-step chooser switched off
    assert b.options[A.OPT_PACKED] == 1 and b._auto is False and c.kernel_switches == 1
    b.redo += 10                                                         # 10 / 800 = 1.25 % of env-steps: tolerated (7 %)
    c._choose_kernel()
    assert b.options[A.OPT_PACKED] == 1 and c.kernel_switches == 1
    b.redo += 80                                                         # 10 %: to the one-env steps
    c._choose_kernel()
    assert b.options[A.OPT_PACKED] == 0 and c.kernel_switches == 2
    b.nefc[3] = 41                                                       # an environment beyond a slot's rows: stays
    c._choose_kernel()
    assert b.options[A.OPT_PACKED] == 0 and c.kernel_switches == 2
    b.nefc[3] = 36                                                       # nobody above 38 rows: back
    c._choose_kernel()
    assert b.options[A.OPT_PACKED] == 1 and c.kernel_switches == 3
    c2 = SegmentCollector.__new__(SegmentCollector)
    c2.env, c2.T, c2.n, c2._redo_seen, c2.kernel_switches = FakeEnv(False), 100, 8, None, 0
    c2._choose_kernel()                                                  # DPVecEnv(packed=False / True), float32, v1-quat reward: not touched
    assert c2.env.batch.options == {} and c2.kernel_switches == 0 and c2.env.batch._auto is True


def test_segment_dict_materialises_pending_episode_lists_on_access():
    """`rollout.Segment`: the generator's segment dict whose "ep_rets" / "ep_lens" may still be on their way from the device (the collector's
    episode scan): they appear on the first read, on iteration, in `in`, in `dict(seg)` — and exactly once; a segment without pending lists
    is a plain dict (KeyError for what is not there)."""
    from deepmimic_mujoco_amd.rollout import Segment

    class Pending:
        calls = 0

        def result(self):
            Pending.calls += 1
            return [1.5, 2.5], [3, 4]

    s = Segment(); s["ob"] = 0; s.pending_episodes = Pending()
    assert "ep_rets" in s and "ep_lens" in s and "nope" not in s and Pending.calls == 0
    assert s["ep_lens"] == [3, 4] and s["ep_rets"] == [1.5, 2.5] and Pending.calls == 1 and s.pending_episodes is None
    s2 = Segment(); s2["ob"] = 0; s2.pending_episodes = Pending()
    assert sorted(s2) == ["ep_lens", "ep_rets", "ob"] and Pending.calls == 2          # whoever walks the dict sees all of it
    s3 = Segment(); s3.pending_episodes = Pending()
    assert dict(s3) == {"ep_rets": [1.5, 2.5], "ep_lens": [3, 4]} and s3.get("ep_rets") == [1.5, 2.5] and s3.get("x", 7) == 7 and len(s3) == 2
    s4 = Segment(); s4.pending_episodes = Pending()
    s4.finish_episode_stats(); s4.finish_episode_stats()
    assert Pending.calls == 4 and s4["ep_rets"] == [1.5, 2.5]
    plain = Segment(ob=1)
    with pytest.raises(KeyError):
        plain["ep_rets"]
    assert "ep_rets" not in plain and list(plain) == ["ob"]
