"""GPU tests of the device-resident rollout (policy + env kernel + GAE on one stream) and the statistical anchor of the
physics on the HIP path: the reference's shipped policy must live as long in the kernel's physics as in its own
training log (real MuJoCo)."""
import os

import numpy as np
import pytest
import torch

from deepmimic_mujoco_amd import _abi as A

from deepmimic_mujoco_amd import DPVecEnv, MlpPolicy, traj_segment_generator, add_vtarg_and_adv
from tests.test_policy import CKPT, GOLD, _gae_reference_loop

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run(pol, stochastic, n, steps, seed=0, first_reset="rsi"):
    env = DPVecEnv(n, motion="walk", device=0, reward="alive", autoreset="init", seed=seed)
    pol.seed(seed)
    gen = traj_segment_generator(pol, env, steps, stochastic=stochastic, first_reset=first_reset)
    seg = next(gen)
    env.close()
    return seg


def test_rollout_segment_is_consistent_and_gae_matches_reference_loop():
    pol = MlpPolicy.from_tf_checkpoint(CKPT, device=DEV)
    T, n = 300, 64
    seg = _run(pol, True, n, T, seed=2)
    assert seg["ob"].shape == (T, n, 56) and seg["ac"].shape == (T, n, 28) and seg["ob"].device.type == "cuda"
    assert bool(torch.isfinite(seg["ob"]).all()) and bool((seg["new"][0] == 1).all())
    assert float(seg["rew"].min()) == 1.0 and float(seg["rew"].max()) == 1.0          # alive reward of dp_env_v3.py:117-128
    # vpred / ac are the policy's outputs on the stored observations (float32 storage of float64 obs costs ~1e-6)
    mean, vpred = pol.forward(seg["ob"].reshape(-1, 56))
    assert float((vpred.reshape(T, n) - seg["vpred"]).abs().max()) < 1e-3 * max(1.0, float(seg["vpred"].abs().max()))
    # episodes: every new[t]=1 (t>0) closes an episode whose length was logged; an episode that ends on the segment's last
    # step is logged here too, but its `new` flag opens the NEXT segment (its nextvpred is zeroed, src/trpo.py:86-88)
    starts = seg["new"].cpu().numpy()
    ends_on_last = int((seg["nextvpred"] == 0).sum())
    assert int(starts[1:].sum()) + ends_on_last == len(seg["ep_lens"]) and len(seg["ep_lens"]) > 0
    assert abs(sum(seg["ep_rets"]) - sum(seg["ep_lens"])) < 1e-9
    add_vtarg_and_adv(seg, 0.995, 0.97)
    rew, vp, nw, nxt = (seg[k].cpu().numpy() for k in ("rew", "vpred", "new", "nextvpred"))
    for e in range(0, n, 7):
        adv, ret = _gae_reference_loop(rew[:, e], vp[:, e], nw[:, e], nxt[e], 0.995, 0.97)
        assert np.allclose(seg["adv"][:, e].cpu().numpy(), adv, rtol=1e-4, atol=1e-3)
        assert np.allclose(seg["tdlamret"][:, e].cpu().numpy(), ret, rtol=1e-4, atol=1e-3)


def test_shipped_policy_episode_lengths_on_gpu_match_training_log():
    log = np.load(os.path.join(GOLD, "trpo_walk0_log.npz"))["EpLenMean"]
    n = 2048
    seg = _run(MlpPolicy(device=DEV, seed=0), True, n, 160, seed=1, first_reset="init")
    first = _first_lengths(seg, 160)
    print("untrained policy: first-episode length mean %.1f (reference log, first 5 iterations: %.1f)" % (first.mean(), log[:5].mean()))
    assert abs(first.mean() - log[:5].mean()) < 4
    seg = _run(MlpPolicy.from_tf_checkpoint(CKPT, device=DEV), True, n, 2000, seed=1, first_reset="init")
    first = _first_lengths(seg, 2000)
    print("shipped policy: first-episode length mean %.1f median %.0f (reference log, last 100 iterations: %.1f, range %.0f-%.0f)"
          % (first.mean(), np.median(first), log[-100:].mean(), log[-100:].min(), log[-100:].max()))
    assert 0.8 * log[-100:].mean() < first.mean() < 1.35 * log[-100:].mean()
    seg = _run(MlpPolicy.from_tf_checkpoint(CKPT, device=DEV), False, 512, 1000, seed=1, first_reset="init")
    alive = float((_first_lengths(seg, 1000) == 1000).mean())
    print("deterministic shipped policy: %.1f%% of envs still up after 1000 steps" % (100 * alive))
    assert alive > 0.9


def _first_lengths(seg, cap):
    new = seg["new"].cpu().numpy()[1:]                       # new[t]=1: the episode ended at step t
    T, n = new.shape
    first = np.where(new.any(0), new.argmax(0) + 1, cap)
    return first


def test_trpo_learns_to_stay_up_on_gpu():
    """The reference's learner loop (src/trpo.py:97-319) on device-resident rollouts: a few seconds of training must lift the
    episode length well above the untrained ~34 steps (the reference needed ~300 k timesteps / ~10 min for the same)."""
    from deepmimic_mujoco_amd.trpo import learn
    env = DPVecEnv(1024, motion="walk", device=0, reward="alive", autoreset="init", seed=0)
    pi = MlpPolicy(device=DEV, seed=0); pi.seed(0)
    hist = learn(env, pi, timesteps_per_batch=64, max_iters=70, vf_batch_size=4096, log=None)
    env.close()
    first = np.mean([h["EpLenMeanIter"] for h in hist[:3]]); last = np.mean([h["EpLenMeanIter"] for h in hist[-5:]])
    print("TRPO: EpLenMean %.1f -> %.1f in %d iterations, %.1f s, %d env steps" % (first, last, len(hist), hist[-1]["TimeElapsed"], hist[-1]["TimestepsSoFar"]))
    assert first < 45 and last > 1.8 * first
    assert all(h["meankl"] <= 0.0151 for h in hist) and all(np.isfinite(h["surrgain"]) for h in hist)


def test_native_policy_kernel_matches_torch_forward():
    """dm_policy_act (one HIP launch) against the torch graph the learner differentiates: means and values to float32 rounding,
    sampled actions = mean + exp(logstd) * standard normal noise, a fresh stream per call and per (env, action)."""
    pol = MlpPolicy.from_tf_checkpoint(CKPT, device=DEV)
    pol.seed(5)
    n = 4096 + 17                                                     # not a multiple of the kernel's 32-env tile
    g = torch.Generator(device=DEV); g.manual_seed(0)
    ob = (pol.ob_rms.mean + pol.ob_rms.std * torch.randn((n, 56), generator=g, device=DEV) * 2.5).to(torch.float64).contiguous()
    mean, vpred = pol.forward(ob)
    ac, vp = pol.act(False, ob)                                       # native path
    assert ac.dtype == torch.float64 and float((ac - mean.to(torch.float64)).abs().max()) < 2e-4 * max(1.0, float(mean.abs().max()))
    assert float((vp - vpred).abs().max()) < 2e-4 * max(1.0, float(vpred.abs().max()))
    pol.native = False
    ac_t, vp_t = pol.act(False, ob)                                   # torch path, same API
    pol.native = True
    assert float((ac - ac_t).abs().max()) < 2e-4 * max(1.0, float(mean.abs().max()))
    a1, _ = pol.act(True, ob); a2, _ = pol.act(True, ob)
    resid = ((a1 - mean.to(torch.float64)) / torch.exp(pol.params["logstd"]).to(torch.float64)).to(torch.float32)
    assert abs(float(resid.mean())) < 0.01 and abs(float(resid.std()) - 1.0) < 0.01 and float(resid.abs().max()) < 6.5
    assert not torch.equal(a1, a2)                                    # the counter advances
    c = torch.corrcoef(torch.stack([resid[:, 0], resid[:, 1]]))[0, 1]
    assert abs(float(c)) < 0.05                                       # independent across actions
    # parameters changed in place are picked up after mark_dirty()
    with torch.no_grad():
        pol.params["polfinal/b"] += 0.5
    pol.mark_dirty()
    ac2, _ = pol.act(False, ob)
    assert float((ac2 - ac - 0.5).abs().max()) < 1e-4


@pytest.mark.gpu
def test_bench_prints_one_contract_line():
    """bench.py's output contract: ONE JSON line with the metric, the roofline object and (N = 1) the CPU baseline."""
    import json
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "16", "--warmup", "4", "--prewarm-horizons", "0",
                          "--envs", "512"], capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 16 and j["warmup"] == 4 and j["dtype"] == "f64" and j["scaling"] == "weak" and j["vs_baseline"] is None
    assert j["value"] > 1e5 and "workload" in j["config"]
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6 and r["peak"] == 8000.0
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 1000 and "sample" in c
    # the default line: dm_batch_step calls queued into one horizon launch per window; three windows; the facade's figure beside it
    assert j["config"]["step_queue"] == 256 and r["kernel"] == "k_rollout_packed" and r["launch"]["steps_per_launch"] == 16 and r["launch"]["avg_us"] > 0
    sp = j["value_spread"]
    assert sp["windows"] == 3 and sp["min"] <= sp["median"] == j["value"] <= sp["max"]
    ve = j["vecenv_step"]
    assert ve["steps"] == 16 and ve["value"] > 1e5 and ve["kernel"] == "k_step_narrow"
    # the headline states its own precondition: the step queue is a non-default option, and what `value` therefore is
    assert j["config"]["non_default_options"] == ["DM_OPT_STEP_QUEUE=256"] and "vecenv_step" in j["config"]["value_is"]
    assert j["config"]["distinct_gpus"] == 1 and j["config"]["rank_devices"] is None        # (single process: no process group to gather the PCI addresses over)


@pytest.mark.gpu
def test_bench_two_ranks_share_the_gpu_over_gloo():
    """The multi-rank code path of bench.py executed on ONE GPU: two processes (torch.distributed.run), env ranges sharded by
    rank, the double-buffered per-horizon rollout gather (host-staged over gloo), MAX-reduced timing, rank 0 prints the line."""
    import json
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29541",
           os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--envs", "512", "--steps", "300", "--warmup", "8",
           "--prewarm-horizons", "0"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["n_ranks_seen"] == 2 and j["config"]["dist_backend"] == "gloo"
    assert j["config"]["global_envs"] == 1024 and j["config"]["gathers_completed"] >= 1 and j["scaling"] == "weak"
    assert j["value"] > 1e5 and "cpu_baseline" not in j
    # the start-up line of a multi-rank run says which GPU every rank sits on (gathered over the process group): two ranks sharing the one GPU here
    assert "rank -> GPU" in out.stderr and "1 distinct GPU(s) for 2 rank(s)" in out.stderr
    assert len(j["config"]["rank_devices"]) == 2 and j["config"]["distinct_gpus"] == 1 and j["config"]["rank_devices"][0].rsplit(".", 1)[0] == j["config"]["rank_devices"][1].rsplit(".", 1)[0]


@pytest.mark.gpu
def test_bench_rccl_code_path_runs_at_world_size_one():
    """The RCCL branch of the multi-GPU path (src/train_mpi.sh:1, src/trpo.py:175-186, src/mpi_adam.py:21-50 in the reference: MPI) executed
    on the one GPU there is: `init_process_group("nccl", device_id=...)`, the double-buffered device-buffer `all_gather_into_tensor` of the
    [256, 8192, 87] f32 rollout block of configs[4]'s shard, the barriers and the max-reduced timing on device tensors, the learner's all-mean
    on a gradient-sized vector — with a group of ONE rank, so that no second GPU is needed; and the two 8-way gathered buffers an 8-rank job
    holds per rank (5.8 GB each) allocate beside the batch.  It gives no scaling number; it removes "never ran" from the first SCALE run."""
    import json
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--workload", "cfg5", "--steps", "512", "--warmup", "0", "--prewarm-horizons", "0",
                          "--force-dist", "--dist-backend", "nccl", "--alloc-gather-world", "8", "--repeats", "1", "--no-pmc", "--no-cpu-baseline", "--no-gym-loop",
                          "--no-vecenv-leg", "--no-horizon-leg"], capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "nccl process group up, 1 ranks (backend reports nccl)" in out.stderr
    assert "1 distinct GPU(s) for 1 rank(s)" in out.stderr                                  # the PCI address travelled over RCCL (an all_gather of a device tensor)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{"metric"')]       # (RCCL may print to stdout when the group is torn down)
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    c = j["config"]
    assert c["n_ranks_seen"] == 1 and c["dist_backend"] == "nccl" and c["forced_dist"] is True
    assert c["gathers_completed"] >= 2, "two horizons: both blocks of the double buffer travelled"
    assert c["learner_allmean_ok"] is True
    assert c["gather_buffers_allocated_bytes"] == 2 * 8 * 256 * 8192 * 87 * 4
    assert j["value"] > 1e6 and j["n_gpus"] == 1 and c["envs_per_gpu"] == 8192


@pytest.mark.gpu
def test_bench_eight_ranks_share_the_gpu_over_gloo():
    """World size 8 — the driver's largest scaling point — on the ONE visible GPU: shard offsets 0 .. 7 N, eight-way gathered rollout
    blocks (DoubleBufferedGather, host-staged over gloo), MAX-reduced timing, exactly one JSON line from rank 0.  BASELINE.json
    configs[4]'s clip and reset mode at a reduced shard size (eight processes share one device)."""
    import json
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", "29543",
           os.path.join(root, "bench.py"), "--gpus", "8", "--dist-backend", "gloo", "--workload", "cfg5", "--envs", "256", "--steps", "300",
           "--warmup", "8", "--prewarm-horizons", "0"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["config"]["n_ranks_seen"] == 8 and j["config"]["dist_backend"] == "gloo"
    assert j["config"]["global_envs"] == 8 * 256 and j["config"]["gathers_completed"] >= 1 and j["scaling"] == "weak"
    assert j["config"]["clip"] == "dance_b" and j["value"] > 1e4 and "cpu_baseline" not in j
    assert "rollout gather" in out.stderr, "the multi-rank start-up line (rank count, gather size) is missing"
    # what a SCALE record is read for (round 6): every rank reported its device over the process group, all eight sit on the one GPU of this box, and
    # both blocks of the double buffer travelled (300 steps = one whole 256-step horizon + a started one)
    c = j["config"]
    assert len(c["rank_devices"]) == 8 and len(set(d.rsplit(".", 1)[0] for d in c["rank_devices"])) == 1
    assert c["distinct_gpus"] == 1, "eight ranks on one visible device: %r" % (c["distinct_gpus"],)
    assert "1 distinct GPU(s) for 8 rank(s)" in out.stderr
    assert c["gathers_completed"] >= 1 and c["envs_per_gpu"] == 256


@pytest.mark.gpu
def test_two_rank_training_keeps_replicas_bit_identical():
    """tools/train_trpo.py on two ranks (gloo, sharing the GPU): after three TRPO updates — all-mean'd policy gradient, Fisher-vector
    products, value gradients, all-reduced filter moments — both replicas hold bit-identical parameters, although their env
    shards (and hence their rollouts) differ.  The reference's consistency check: src/mpi_adam.py:42-50."""
    import subprocess
    import sys
    import tempfile
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    with tempfile.TemporaryDirectory() as td:
        pre = os.path.join(td, "p")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29545",
               os.path.join(root, "tools", "train_trpo.py"), "--dist-backend", "gloo", "--envs", "256", "--horizon", "32", "--iters", "3",
               "--vf-batch", "1024", "--dump-params", pre]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root)
        assert out.returncode == 0, out.stderr[-3000:]
        a = np.load(pre + ".rank0.npz"); b = np.load(pre + ".rank1.npz")
        assert sorted(a.files) == sorted(b.files) and len(a.files) >= 14
        for k in a.files:
            assert np.array_equal(a[k], b[k]), "replicas diverged in %s" % k
        fresh = MlpPolicy(device="cpu", seed=0).state_dict()
        moved = [k for k in a.files if k in fresh and not np.array_equal(a[k], np.asarray(fresh[k]))]
        assert len(moved) >= 10, "three updates must have changed the parameters (%s)" % moved


@pytest.mark.gpu
def test_pipelined_rollouts_equal_the_same_batches_stepped_one_after_the_other():
    """`pipelined_segment_generator`: two env batches whose policy -> env chains run concurrently on their own CUDA streams must
    produce exactly the segments the same two batches produce when their chains run back to back on one stream."""
    from deepmimic_mujoco_amd import SegmentCollector, pipelined_segment_generator
    T, n = 40, 192
    outs = []
    for piped in (True, False):
        envs = [DPVecEnv(n, motion="walk", device=0, reward="alive", autoreset="init", seed=4, env_offset=h * n) for h in range(2)]
        pol = MlpPolicy.from_tf_checkpoint(CKPT, device=DEV); pol.seed(9)
        if piped:
            gen = pipelined_segment_generator(pol, envs, T, stochastic=True)
            segs = [next(gen) for _ in range(3)]
        else:
            cols = [SegmentCollector(pol, e, T, True, None, "rsi") for e in envs]
            segs = []
            for _ in range(3):
                for c in cols:
                    c.launch()
                parts = [c.collect() for c in cols]
                segs.append({k: (torch.cat([p[k] for p in parts], 0 if k == "nextvpred" else 1) if k not in ("ep_rets", "ep_lens")
                                 else [x for p in parts for x in p[k]]) for k in parts[0]})
        torch.cuda.synchronize()
        outs.append(segs)
        for e in envs:
            e.close()
    for a, b in zip(*outs):
        assert a["ob"].shape == (T, 2 * n, 56) and a["nextvpred"].shape == (2 * n,)
        for k in ("ob", "ac", "rew", "vpred", "new", "prevac", "nextvpred"):
            assert torch.equal(a[k], b[k]), k
        assert a["ep_lens"] == b["ep_lens"] and a["ep_rets"] == b["ep_rets"]
    assert sum(len(s["ep_lens"]) for s in outs[0]) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [["--workload", "cfg4"], ["--workload", "cfg5"], ["--workload", "cfg2", "--pipeline", "1"], ["--dtype", "32"], ["--horizon-launch"],
                                   ["--step-queue", "0"], ["--step-queue", "8", "--repeats", "1"]])
def test_bench_other_workloads_print_the_contract_line(extra):
    """The single-shard lines of BASELINE.json configs[3] / [4] ('spinkick', 'dance_b'; an interior shard's global env ids), configs[1]
    and the float32 build go through the same code path and print the same contract line."""
    import json
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "24", "--warmup", "4", "--prewarm-horizons", "0", "--envs", "768",
                          "--no-pmc", "--no-cpu-baseline", "--no-gym-loop"] + extra, capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads([ln for ln in out.stdout.splitlines() if ln.strip()][-1])
    assert j["value"] > 1e5 and j["n_gpus"] == 1 and j["steps"] == 24 and j["config"]["envs_per_gpu"] == 768
    wl = extra[1] if extra[0] == "--workload" else "cfg3"
    assert {"cfg4": "spinkick", "cfg5": "dance_b", "cfg2": "walk", "cfg3": "walk"}[wl] == j["config"]["clip"]
    if wl in ("cfg4", "cfg5"):
        assert "shard 3 of 8" in j["config"]["workload"] and j["config"]["global_envs"] == 8 * 768
    assert j["dtype"] == ("f32" if "--dtype" in extra else "f64")
    if "--horizon-launch" in extra:          # the timed leg itself goes through dm_batch_rollout: one launch for the 24 steps
        assert j["roofline"]["kernel"] == "k_rollout_packed" and j["roofline"]["launch"]["steps_per_launch"] == 24 and j["horizon_launch"] is None
        return
    queued = full_queue = wl != "cfg2" and "--dtype" not in extra and extra[:2] != ["--step-queue", "0"]
    assert bool(j["config"]["step_queue"]) == queued
    if queued:                               # the timed dm_batch_step calls ran as horizon launches of the queue's depth
        depth = 8 if "8" in extra else 24
        assert j["roofline"]["kernel"] == "k_rollout_packed" and j["roofline"]["launch"]["steps_per_launch"] == depth
        assert j["config"]["step_queue_launches"] >= 24 // depth
    else:
        assert j["roofline"]["launch"]["launches_per_step"] == (1 if "--pipeline" in extra else 2)
    assert j["value_spread"]["windows"] == (1 if "--repeats" in extra else 3)
    assert (j["vecenv_step"] is None) == ("--dtype" in extra)
    hl = j["horizon_launch"]                 # the second leg: whole 256-step horizons of the same workload through dm_batch_rollout
    assert (hl is None) == ("--dtype" in extra)
    if hl is not None:
        assert hl["steps"] == 256 and hl["value"] > 1e5 and hl["unit"] == "env-steps/s"


@pytest.mark.gpu
@pytest.mark.parametrize("pipeline,packed", [(1, False), (2, False), (1, True), (2, True)])
def test_fused_policy_step_equals_step_then_act(pipeline, packed):
    """dm_batch_step_act == dm_batch_step followed by dm_policy_act on the observations it produced: same obs / reward / done
    (bit-exact: the same env kernel code), same actions and values (the per-wave MLP sums in a different order: 1e-5), over
    closed-loop steps with auto-resets (an untrained policy: episodes of ~34 steps), at one launch per step and with pipelined
    sub-batches; `packed`: the same on the four-environments-per-wavefront kernel (k_step_packed_act: one weight stream per wave
    serves four environments; 642 envs = a last wave with spare slots)."""
    from deepmimic_mujoco_amd import _abi as A
    n, steps = (642 if packed else 640), 48
    pol = MlpPolicy(device=DEV, seed=2); pol.seed(5)
    outs = []
    for fused in (False, True):
        env = DPVecEnv(n, motion="walk", device=0, reward="alive", autoreset="init", seed=3, packed=packed)
        env.batch.set_option(A.OPT_PIPELINE, pipeline)
        ob = torch.zeros((steps + 1, n, 56), dtype=torch.float64, device=DEV)
        ac = torch.zeros((steps + 1, n, 28), dtype=torch.float64, device=DEV)
        vp = torch.zeros((steps + 1, n), dtype=torch.float32, device=DEV)
        rew = torch.zeros((steps, n), dtype=torch.float64, device=DEV); dn = torch.zeros((steps, n), dtype=torch.uint8, device=DEV)
        env.reset("init", out=ob[0])
        pol._counter = 100
        pol.act(True, ob[0], out=ac[0], vpred_out=vp[0])
        for t in range(steps):
            if fused:
                pol._counter += 1
                env.batch.step_act(ac[t], 1, (ob[t + 1], rew[t], dn[t]), pol._packed, ac[t + 1], vp[t + 1], True, pol._seed, pol._counter)
            else:
                env.batch.step(ac[t], 1, (ob[t + 1], rew[t], dn[t]))
                env.batch.join()
                pol.act(True, ob[t + 1], out=ac[t + 1], vpred_out=vp[t + 1])
        env.batch.sync()
        outs.append((ob.clone(), ac.clone(), vp.clone(), rew.clone(), dn.clone()))
        env.close()
    (ob0, ac0, vp0, r0, d0), (ob1, ac1, vp1, r1, d1) = outs
    # the first step is bit-identical (same actions in); its policy outputs differ by float32 rounding; later steps inherit that
    assert torch.equal(ob0[1], ob1[1]) and torch.equal(d0[0], d1[0])
    assert float((ac0[1] - ac1[1]).abs().max()) < 1e-5 and float((vp0[1] - vp1[1]).abs().max()) < 1e-4 * max(1.0, float(vp0[1].abs().max()))
    assert float((ob0[:9] - ob1[:9]).abs().max()) < 1e-4 and float((ac0[:9] - ac1[:9]).abs().max()) < 1e-4
    same = (d0 == d1).all(0)                                  # envs whose episodes ended at the same steps in both runs
    assert float(same.float().mean()) > 0.9 and int(d0.sum()) > n // 2 and abs(int(d0.sum()) - int(d1.sum())) <= n // 20
    # an auto-reset env's next action is the policy's on the FRESH episode's observation
    t, e = [int(x) for x in d1.nonzero()[0]]
    assert float(ob1[t + 1, e, :28].abs().max()) < 0.0101
    ref_ac, ref_vp = pol.forward(ob1[t + 1, e][None])
    assert abs(float(ref_vp[0]) - float(vp1[t + 1, e])) < 1e-4 * max(1.0, abs(float(ref_vp[0])))


@pytest.mark.gpu
def test_fused_segment_generator_follows_the_reference_protocol():
    """`traj_segment_generator(fused=True)`: segments of the same shape and statistics as the two-launch form; the first action of a
    segment is the one drawn for that observation at the end of the previous segment (src/trpo.py:47-56), `new` / `prevac` / episode
    bookkeeping hold across segment boundaries, and the shipped policy balances as long as in the two-launch form."""
    from deepmimic_mujoco_amd import _abi as A
    n, T = 512, 64
    pol = MlpPolicy.from_tf_checkpoint(CKPT, device=DEV); pol.seed(1)
    env = DPVecEnv(n, motion="walk", device=0, reward="alive", autoreset="init", seed=0)
    env.batch.set_option(A.OPT_PIPELINE, 2)
    gen = traj_segment_generator(pol, env, T, stochastic=True, first_reset="init", fused=True)
    segs = [next(gen) for _ in range(8)]
    lens = [x for s in segs for x in s["ep_lens"]]
    for a, b in zip(segs[:-1], segs[1:]):
        assert torch.equal(b["prevac"][0], a["ac"][-1])                     # prevac of row 0 = last action of the previous segment
        assert a["ob"].shape == (T, n, 56) and a["ac"].shape == (T, n, 28) and a["vpred"].shape == (T, n)
    for s in segs:
        assert bool((s["rew"] == 1).all()) and bool(torch.isfinite(s["ac"]).all()) and bool(torch.isfinite(s["vpred"]).all())
        fresh = s["ob"][1:][s["new"][1:].bool()]
        assert fresh.numel() == 0 or bool((fresh[:, :28].abs() < 0.0101).all())      # a fresh episode starts at the noisy default pose
    assert len(lens) > 50 and 150 < np.mean(lens) < 450                     # the checkpoint's policy: ~270 steps
    env.close()


@pytest.mark.gpu
def test_evaluate_task_runs_the_shipped_checkpoint():
    """`tools/train_trpo.py --task evaluate --load-model-path <the reference's checkpoint>` = src/trpo.py:480-487 (`runner`): prints the
    reference's three lines; the shipped policy stays up for hundreds of steps (its own log: 264 on average, stochastic)."""
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "train_trpo.py"), "--task", "evaluate", "--load-model-path", CKPT,
                          "--number-trajs", "64", "--stochastic-policy"], capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert lines[-3] == "stochastic policy:" and lines[-2].startswith("Average length:") and lines[-1].startswith("Average return:")
    avg = float(lines[-2].split(":")[1])
    assert 150 < avg < 500 and abs(float(lines[-1].split(":")[1]) - avg) < 1e-6          # alive reward: return == length


@pytest.mark.gpu
@pytest.mark.parametrize("n,pipeline", [(1, 2), (3, 2), (65, 1)])
def test_fused_policy_step_on_tiny_and_odd_batches(n, pipeline):
    """dm_batch_step_act with fewer envs than sub-batches / odd sizes: same observations as dm_batch_step (bit-exact) and the policy
    kernel's actions / values on them."""
    from deepmimic_mujoco_amd import _abi as A
    pol = MlpPolicy(device=DEV, seed=4); pol.seed(4)
    res = []
    for fused in (False, True):
        env = DPVecEnv(n, motion="walk", device=0, reward="alive", autoreset="init", seed=8)
        env.batch.set_option(A.OPT_PIPELINE, pipeline)
        ob0 = torch.zeros((n, 56), dtype=torch.float64, device=DEV); env.reset("init", out=ob0)
        ac0 = torch.zeros((n, 28), dtype=torch.float64, device=DEV); vp0 = torch.zeros(n, dtype=torch.float32, device=DEV)
        pol._counter = 10; pol.act(True, ob0, out=ac0, vpred_out=vp0)
        ob1 = torch.zeros_like(ob0); rew = torch.zeros(n, dtype=torch.float64, device=DEV); dn = torch.zeros(n, dtype=torch.uint8, device=DEV)
        ac1 = torch.zeros_like(ac0); vp1 = torch.zeros_like(vp0)
        if fused:
            pol._counter += 1
            env.batch.step_act(ac0, 1, (ob1, rew, dn), pol._packed, ac1, vp1, True, pol._seed, pol._counter)
        else:
            env.batch.step(ac0, 1, (ob1, rew, dn)); env.batch.join()
            pol.act(True, ob1, out=ac1, vpred_out=vp1)
        env.batch.sync()
        res.append((ob1.clone(), ac1.clone(), vp1.clone()))
        env.close()
    assert torch.equal(res[0][0], res[1][0])
    assert float((res[0][1] - res[1][1]).abs().max()) < 1e-5 and float((res[0][2] - res[1][2]).abs().max()) < 1e-4


@pytest.mark.gpu
def test_auto_packed_follows_the_workload():
    """DPVecEnv(packed=None) at 8192 envs starts four-per-wave and re-decides from the batch's own row statistics: RSI + random actions
    (the benchmark regime: envs fall, few rows) stays on the lean packed kernel; a population standing on both feet (the init pose under zero
    actions: 8 foot corners x 4 pyramid rows = 32 rows, 33 .. 37 with joint limits) overflows the 32-row capacity of the lean per-step launch, and
    — its overflows being all ROW overflows — is moved to the per-step launches with the three-set code (OPT_PACKED 2: 40 rows per env; rounds 3-4
    handed it to the one-env kernel), where next to nothing is re-stepped any more; once nobody holds more than 32 rows (RSI + random actions
    again) the batch returns to the lean kernel.  The hand-over to the one-env kernel remains for overflows of another kind: driven here
    through its threshold."""
    n = 8192
    env = DPVecEnv(n, motion="walk", device=0, reward="alive", autoreset="rsi", seed=1)
    b = env.batch
    assert env.packed and b._auto and b.options[A.OPT_PACKED] == 1
    b.ADAPT_EVERY = 32
    env.reset("rsi")
    g = torch.Generator(device=DEV); g.manual_seed(0)
    for t in range(96):
        env.step(torch.randn((n, 28), generator=g, device=DEV, dtype=torch.float64) * 0.9)
    assert b.options[A.OPT_PACKED] == 1 and b.auto_switches == 0, "the benchmark regime must stay on the lean packed kernel"
    env.reset("qpos0")                                               # everybody upright on both feet
    zero = torch.zeros((n, 28), device=DEV, dtype=torch.float64)
    b.set_option(A.OPT_AUTORESET, 2)                                 # fallen envs restart upright
    for t in range(64):
        env.step(zero)
    assert b.options[A.OPT_PACKED] == 2 and b.auto_switches == 1, "a standing population moves to the three-set per-step kernel (redo %s)" % (b.redo_reasons(),)
    r0 = b.redo_total()
    for t in range(64):
        env.step(zero)
    assert b.options[A.OPT_PACKED] == 2 and b.auto_switches == 1 and b.redo_total() - r0 < 3e-4 * 64 * n, b.redo_reasons()
    assert int((b.get(A.F_NEFC) > 32).sum()) > 0
    b.set_option(A.OPT_AUTORESET, 1)
    env.reset("rsi")
    for t in range(128):
        env.step(torch.randn((n, 28), generator=g, device=DEV, dtype=torch.float64) * 0.9)
    assert b.options[A.OPT_PACKED] == 1 and b.auto_switches == 2
    b.REDO_RATE_MAX = -1.0                                           # any redo rate is too much, and (no redo at all: share 0) not for rows: the one-env kernel
    for t in range(32):
        env.step(zero)
    assert not env.packed and b.auto_switches == 3
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("packed,policy", [(True, False), (True, True), (False, True)])
def test_horizon_launch_equals_step_by_step(packed, policy):
    """dm_batch_rollout == T dm_batch_step / dm_batch_step_act calls, bit for bit: observations, rewards, done flags, (with the policy in
    the loop) actions and values of every step, and the final state.  On the packed path the horizon is ONE launch (k_rollout_packed: every
    wavefront runs its four environments through all T steps at its own pace; two environments are planted in poses with more constraint
    rows than a slot holds, so the in-wave re-step by the one-env code runs too); on the one-env path the library issues the T launches."""
    from tests import helpers as H
    n, T = 642, 40
    hi, hq, hv = H.many_row_states(40, 64, want=2)
    pol = MlpPolicy(device=DEV, seed=2); pol.seed(5)
    outs = []
    for horizon in (False, True):
        env = DPVecEnv(n, motion="walk", device=0, reward="v3-config", autoreset="rsi", seed=3, packed=packed)
        env.batch.set_option(A.OPT_PIPELINE, 2)
        b = env.batch
        ob = torch.zeros((T + 1, n, 56), dtype=torch.float64, device=DEV)
        g = torch.Generator(device=DEV); g.manual_seed(7)
        ac = torch.randn((T + 1, n, 28), generator=g, dtype=torch.float64, device=DEV) * 0.9
        vp = torch.zeros((T + 1, n), dtype=torch.float32, device=DEV)
        rew = torch.zeros((T, n), dtype=torch.float64, device=DEV); dn = torch.zeros((T, n), dtype=torch.uint8, device=DEV)
        env.reset("rsi", out=ob[0])
        q = b.get(A.F_QPOS); v = b.get(A.F_QVEL); f = b.get(A.F_FRAME_IDX)
        if not policy:
            for e, k in ((1, 0), (n - 3, -1)):
                q[e], v[e], f[e] = hq[k], hv[k], hi[k]
            b.set_state(q, v, f)
            b.get_obs(ob[0])
        redo0 = b.redo_total()
        pol._counter = 100
        if policy:
            pol.act(True, ob[0], out=ac[0], vpred_out=vp[0])
        if horizon:
            b.rollout(ac, (ob[1:], rew, dn), 1, pol._packed if policy else None, vp[1:] if policy else None, True, pol._seed, pol._counter + 1)
        else:
            for t in range(T):
                if policy:
                    b.step_act(ac[t], 1, (ob[t + 1], rew[t], dn[t]), pol._packed, ac[t + 1], vp[t + 1], True, pol._seed, pol._counter + 1 + t)
                else:
                    b.step(ac[t], 1, (ob[t + 1], rew[t], dn[t]))
        b.join(); b.sync()
        outs.append((ob.clone(), ac.clone(), vp.clone(), rew.clone(), dn.clone(), b.get(A.F_QPOS), b.get(A.F_QVEL), b.get(A.F_QACC_WARMSTART),
                     b.get(A.F_FRAME_IDX), b.get(A.F_EPISODE), b.get(A.F_TIME), b.redo_total() - redo0))
        env.close()
    x, y = outs
    assert bool(torch.isfinite(y[0]).all()) and int(y[4].sum()) > n // 4            # early terminations + auto-resets inside the horizon
    if packed and policy and (x[11] or y[11]):
        # an environment past the packed path's capacities gets its policy step from the one-env code's epilogue (per-step: k_step_redo) or
        # from its wave's four-environment epilogue (horizon launch): the same MLP summed in a different order — float32 rounding, which a
        # closed loop then amplifies.  Everything before the first such event is still identical.
        assert float((x[0][:4] - y[0][:4]).abs().max()) < 1e-4 and float((x[1][:4] - y[1][:4]).abs().max()) < 1e-4
        assert float((x[4] == y[4]).all(0).float().mean()) > 0.9
        return
    for i in range(5):
        assert torch.equal(x[i], y[i]), "row arrays differ (%d)" % i
    for i in range(5, 11):
        assert np.array_equal(x[i], y[i]), "final state differs (%d)" % i
    if packed and not policy:
        assert x[11] == y[11] > 0, "capacity overflows: %d per-step, %d in the horizon launch" % (x[11], y[11])


@pytest.mark.gpu
def test_horizon_launch_forms_agree_with_substeps_and_odd_sizes():
    """dm_batch_rollout's two forms on the packed path (DM option 106: 1 = one launch per horizon, 0 = the library issues the step launches)
    with n_substeps = 2 and an env count that leaves spare slots in the last wave: identical rows and final state; with the option left at
    its default the library picks the one-launch form for this batch (constraint rows, at most two packed waves per SIMD)."""
    n, T = 301, 12
    outs = []
    for mode in (0, 1, -1):
        env = DPVecEnv(n, motion="spinkick", device=0, reward="alive", autoreset="rsi", seed=4, packed=True)
        b = env.batch
        b.set_option(106, mode)
        g = torch.Generator(device=DEV); g.manual_seed(3)
        ac = torch.randn((T + 1, n, 28), generator=g, dtype=torch.float64, device=DEV) * 0.9
        ob = torch.zeros((T, n, 56), dtype=torch.float64, device=DEV); rew = torch.zeros((T, n), dtype=torch.float64, device=DEV)
        dn = torch.zeros((T, n), dtype=torch.uint8, device=DEV)
        env.reset("rsi")
        b.rollout(ac, (ob, rew, dn), 2)
        b.join(); b.sync()
        outs.append((ob.clone(), rew.clone(), dn.clone(), torch.as_tensor(b.get(A.F_QPOS)), torch.as_tensor(b.get(A.F_TIME)), torch.as_tensor(b.get(A.F_EPISODE))))
        env.close()
    for k in (1, 2):
        for x, y in zip(outs[0], outs[k]):
            assert torch.equal(x, y)
    assert bool(torch.isfinite(outs[0][0]).all()) and float(outs[0][4].max()) > 0


@pytest.mark.gpu
def test_segment_collector_chooses_the_kernel_from_its_own_horizons():
    """`SegmentCollector` (fused) steps a horizon through dm_batch_rollout and decides, horizon by horizon, between four environments per
    wavefront (one launch per horizon) and one (step launches) from the last horizon's own statistics.  Both an untrained policy's falling
    population and the reference's shipped policy (standing on both feet: 32 .. 37 rows, within the packed path's 40 since round 5) stay on the
    packed horizon launch; with the tolerated overflow rate set below zero — no population produces less — the batch is handed to the one-env
    steps after its first horizon.
    Either way the segments follow the generator's protocol."""
    from deepmimic_mujoco_amd.rollout import SegmentCollector
    n, T = 512, 64
    for shipped, tolerate in ((False, None), (True, None), (True, -1.0)):
        pol = MlpPolicy.from_tf_checkpoint(CKPT, device=DEV) if shipped else MlpPolicy(device=DEV, seed=1)
        pol.seed(2)
        env = DPVecEnv(n, motion="walk", device=0, reward="alive", autoreset="init", seed=0)
        assert env.horizon_packed_ok
        c = SegmentCollector(pol, env, T, stochastic=True, first_reset="init", fused=True)
        if tolerate is not None:
            c.HORIZON_REDO_RATE_MAX = tolerate
        segs = []
        for _ in range(5):
            c.launch(); segs.append(c.collect())
        for a, bseg in zip(segs[:-1], segs[1:]):
            assert torch.equal(bseg["prevac"][0], a["ac"][-1])
        assert all(bool(torch.isfinite(sg["ob"]).all()) and bool((sg["rew"] == 1).all()) for sg in segs)
        if tolerate is None:
            assert c._packed_now and c.kernel_switches == 1, "stays on the packed horizon launch (switched on once, at the first horizon)"
            assert not env.packed, "the collector's choice is scoped to its rollout calls: per-step callers of the env keep its own kernel"
        else:
            assert c.kernel_switches >= 2, "handed to the one-env steps after the first horizon (and back only when no env holds more than HEAVY_ROWS rows)"
        env.close()


@pytest.mark.gpu
def test_kernels_and_launch_forms_hand_the_state_over_to_each_other():
    """One trajectory stepped by a MIX of everything that can step a batch — one-env steps (which park the kinematics of the state they leave
    behind for the next step), a packed horizon launch, per-step packed launches, one-env steps again — against the same trajectory on the
    one-env kernel only: 5-term imitation reward, frame cursors, cycle and episode counters included.  The kernels agree to rounding
    (different summation orders), so the comparison is at 1e-9 over a short run."""
    n = 203
    g = torch.Generator(device=DEV); g.manual_seed(11)
    ac = torch.randn((13, n, 28), generator=g, dtype=torch.float64, device=DEV) * 0.6
    outs = []
    for mixed in (False, True):
        env = DPVecEnv(n, motion="walk", device=0, reward="imitation", autoreset="rsi", seed=8, packed=False)
        b = env.batch
        b.set_option(106, 1)
        ob = torch.zeros((12, n, 56), dtype=torch.float64, device=DEV); rew = torch.zeros((12, n), dtype=torch.float64, device=DEV)
        dn = torch.zeros((12, n), dtype=torch.uint8, device=DEV)
        env.reset("rsi")
        plan = [("one", 0, 3), ("horizon", 3, 5), ("packed", 8, 2), ("one", 10, 2)] if mixed else [("one", 0, 12)]
        for kind, t0, cnt in plan:
            b.set_option(A.OPT_PACKED, 0 if kind == "one" else 1)
            if kind == "horizon":
                b.rollout(ac[t0:t0 + cnt + 1], (ob[t0:t0 + cnt], rew[t0:t0 + cnt], dn[t0:t0 + cnt]), 1)
            else:
                for t in range(t0, t0 + cnt):
                    b.step(ac[t], 1, (ob[t], rew[t], dn[t]))
        b.join(); b.sync()
        outs.append((ob.clone(), rew.clone(), dn.clone(), b.get(A.F_QPOS), b.get(A.F_FRAME_IDX), b.get(A.F_CYCLE), b.get(A.F_EPISODE)))
        env.close()
    x, y = outs
    assert torch.equal(x[2], y[2]) and np.array_equal(x[4], y[4]) and np.array_equal(x[5], y[5]) and np.array_equal(x[6], y[6])
    assert float((x[0] - y[0]).abs().max()) < 1e-9 * max(1.0, float(x[0].abs().max())) and float((x[1] - y[1]).abs().max()) < 1e-9
    assert float(np.abs(x[3] - y[3]).max()) < 1e-9 and bool(torch.isfinite(y[0]).all())


@pytest.mark.gpu
def test_episode_scan_kernel_equals_the_generators_bookkeeping_loop():
    """dm_episode_scan (SegmentCollector._episodes_native) against `cur_ep_ret += rew; cur_ep_len += 1; if new: append, reset` of
    src/trpo.py:68-79 run env by env on the host, over three consecutive segments (episodes that span segments, several ends per env
    and segment, envs that never end): the same returns (float64 sums in step order: exact), lengths and time-major order — and the torch
    formulation it replaces on the device agrees to rounding."""
    import torch
    from deepmimic_mujoco_amd.rollout import SegmentCollector
    T, n = 37, 301
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(5)

    def fresh():
        c = SegmentCollector.__new__(SegmentCollector)
        c.T, c.n, c.device = T, n, dev
        c.cur_ret = torch.zeros(n, dtype=torch.float64, device=dev); c.cur_len = torch.zeros(n, dtype=torch.int64, device=dev)
        c.step_idx = torch.arange(1, T + 1, device=dev, dtype=torch.int64)[:, None]
        return c
    a, b = fresh(), fresh()
    a.EP_HEAD = 64                                                            # (most segments below end more episodes than the first copy holds)
    ret = np.zeros(n); ln = np.zeros(n, dtype=np.int64)
    for seg in range(3):
        rew = rng.randn(T, n)
        done = (rng.rand(T, n) < 0.04).astype(np.uint8)
        done[:, :7] = 0                                                       # envs that never end
        done[:, 7] = 1                                                        # ... and one that ends every step
        want_r, want_l = [], []
        for t in range(T):
            for e in range(n):
                ret[e] += rew[t, e]; ln[e] += 1
                if done[t, e]:
                    want_r.append(ret[e]); want_l.append(int(ln[e])); ret[e] = 0.0; ln[e] = 0
        rew_t, done_t = torch.as_tensor(rew, device=dev), torch.as_tensor(done, device=dev)
        got_r, got_l = a._episodes_native(rew_t, done_t).result()
        assert got_l == want_l and got_r == want_r
        assert np.array_equal(a.cur_len.cpu().numpy(), ln) and np.array_equal(a.cur_ret.cpu().numpy(), ret)
        old_r, old_l = b._episodes_torch(rew_t, done_t.to(torch.bool), T, n, dev)
        assert old_l == want_l and np.allclose(old_r, want_r, rtol=0, atol=1e-9)
    empty_r, empty_l = a._episodes_native(torch.zeros((T, n), dtype=torch.float64, device=dev), torch.zeros((T, n), dtype=torch.uint8, device=dev)).result()
    assert empty_r == [] and empty_l == [] and np.array_equal(a.cur_len.cpu().numpy(), ln + T)



@pytest.mark.gpu
def test_two_env_sets_on_two_streams_step_like_each_alone():
    """INTEGRATION.md section A (round 6): a host-side policy drives TWO env sets, each under its own torch stream, so that one set's step overlaps the other's
    drain (tools/two_batch_bench.py: 22 M env-steps/s in aggregate against 13 M for one set).  The VecEnv contract per set is unchanged
    (src/utils/vec_env/__init__.py:26-100): every set's observations, rewards and done flags are bit-identical to the same set stepped alone on the default stream."""
    n, steps = 512, 64
    g = torch.Generator(device=DEV); g.manual_seed(4)
    acts = [torch.randn((steps, n, 28), generator=g, dtype=torch.float64, device=DEV) * 0.9 for _ in range(2)]

    def make(seed):
        e = DPVecEnv(n, motion="walk", device=0, reward="imitation", autoreset="rsi", seed=seed, packed=True, frame_skip=1)
        e.reset("rsi")
        return e

    def outs():
        return [(torch.empty((steps, n, 56), dtype=torch.float64, device=DEV), torch.empty((steps, n), dtype=torch.float64, device=DEV),
                 torch.empty((steps, n), dtype=torch.uint8, device=DEV)) for _ in range(2)]
    # alone, one after the other, on the current stream
    ref = outs()
    for i in range(2):
        e = make(10 + i)
        for t in range(steps):
            e.step(acts[i][t], out=(ref[i][0][t], ref[i][1][t], ref[i][2][t]))
        e.batch.join(); torch.cuda.synchronize(); e.close()
    # interleaved, a stream each
    got = outs()
    envs = [make(10), make(11)]
    sts = [torch.cuda.Stream(device=DEV) for _ in envs]
    torch.cuda.synchronize()
    for t in range(steps):
        for i, e in enumerate(envs):
            with torch.cuda.stream(sts[i]):
                e.step(acts[i][t], out=(got[i][0][t], got[i][1][t], got[i][2][t]))
    for i, e in enumerate(envs):
        with torch.cuda.stream(sts[i]):
            e.batch.join()
    torch.cuda.synchronize()
    for i in range(2):
        for a, b in zip(ref[i], got[i]):
            assert torch.equal(a, b), "set %d differs between the two-stream and the stand-alone run" % i
        assert int(ref[i][2].sum()) > 0, "the run must contain auto-resets"
    for e in envs:
        e.close()


@pytest.mark.gpu
def test_default_kernel_choice_by_batch_size():
    """DPVecEnv(packed=None) starts four-per-wave from PACKED_FROM_ENVS = 4 096 environments (round 6: 13.3 against 12.4 M env-steps/s closed loop there, 10.3 against
    10.6 M at 3 072 — profiles/r06_ab_kernel_variants.md section 3) and one env per wave below; contact-free models take the packed kernel from 256 envs."""
    from deepmimic_mujoco_amd.dp_env import PACKED_FROM_ENVS
    assert PACKED_FROM_ENVS == 4096
    for n, contacts, want in ((4096, True, True), (3072, True, False), (512, False, True)):
        env = DPVecEnv(n, motion="walk", device=0, reward="alive", autoreset="rsi", seed=0, contacts=contacts, limits=contacts)
        assert bool(env.packed) == want, (n, contacts, env.packed)
        assert bool(env.batch.__dict__.get("_auto")) == want            # the chooser keeps watching the batch's own row statistics
        env.reset("rsi")
        obs, rew, done, _ = env.step(np.zeros((n, 28)))
        assert np.isfinite(obs).all() and obs.shape == (n, 56)
        env.close()
