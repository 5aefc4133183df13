#!/usr/bin/env python3
"""bench.py — env-steps/s of the batched DeepMimic humanoid step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

Workloads (`--workload`, named in config.workload):
  cfg3 (default; the configuration the metric is quoted on) — BASELINE.json configs[2]: 'walk' clip, 4096 envs per GPU, full
        contact + joint-limit solve (PGS 50), the 5-term DeepMimic imitation reward (`--reward v3-config` selects dp_env_v3's
        own disabled config reward instead), RSI auto-reset on done, actions ~ N(0, 0.9^2) i.i.d. (pre-generated on the device).
  cfg2 — configs[1]: 'walk', 4096 envs, P-controller torque, contacts and limits off.
  cfg4 — configs[3]: 'spinkick', 32768 envs sharded 8 x 4096; a single process times ONE shard (shard 3: env_offset = 3 N).
  cfg5 — configs[4]: 'dance_b', reference-state-init + early termination, 65536 envs sharded 8 x 8192; one shard as above.
  rollout — informational: policy in the loop + GAE.
  standing — informational: the reference's shipped policy in the loop (the regime a trained population lives in), three kernel legs.
One "step" = one `dm_batch_step` call = one DPEnv.step (ONE RK4 mj_step, h = 0.0166 s, as src/dp_env_v3.py:108-112 hard-codes) of every env of
the rank.  The timed calls run under DM_OPT_STEP_QUEUE (`--step-queue`, default 256; include/dmenv.h): they are queued and executed together as
one horizon launch per 256 calls or at the join that ends a window — every wavefront steps its four environments through the queued steps at its
own pace instead of waiting for the slowest wave of every step; open-loop stepping with pre-drawn actions, the same contract as the pipelined
sub-batches of rounds 2-3 (outputs valid after `dm_batch_join`), results bit-identical to unqueued steps (tests/test_gpu_queue.py).  THREE timed
windows of `--steps` steps each, every one bracketed by barrier + synchronize (`--repeats`); `value` / `ms_per_step` are the median window,
`value_spread` min / median / max.  With N > 1 the env index range is sharded over ranks (weak scaling, no per-step collective) and every 256
steps the [256, n, 87] f32 rollout block is all-gathered (RCCL with `--dist-backend nccl`, the default; `gloo` stages the block through pinned host
memory so that the multi-rank code path can run where only one GPU is visible; `--force-dist` takes that path with ONE rank) — asynchronously,
double-buffered; every gather completes inside the timed region.  Prints ONE JSON line (rank 0).

Further legs in the same line (N = 1 or max over ranks, same barriers):
  `vecenv_step`     the same number of steps through the facade `DPVecEnv.step(actions, out=...)` with nothing queued: one launch set per call on the
                    kernel DPVecEnv(packed=None) picks — what a caller gets that consumes every step's outputs (the drop-in for VecEnv.step);
  `horizon_launch`  the window rounded up to whole 256-step horizons through `dm_batch_rollout` (T steps per call).
`--step-queue 0` times one launch set per `dm_batch_step` call (rounds 1-3's `value`); `--horizon-launch` makes the rollout call the timed leg.

Besides the contract fields the line carries (N = 1): `roofline` (HBM fraction from the algorithmic bytes per launch and the launch's duration by HIP
events the library records around it; fp64 fraction from a flop count of the kernel's algorithm on the run's own row / sweep statistics; with
rocprofv3 on PATH, live PMC passes of this very workload: HBM traffic, VALU issue fraction, lane efficiency), `cpu_baseline` (the oracle on the host
cores), `single_env_gym_loop` (steps/s of a Python `DPEnv.step` loop, the path src/trpo.py:47-80 drives).
"""
import argparse
import json
import os
import subprocess
import sys
import time
import warnings

import numpy as np

# Multi-process GPU runs on this platform need dmabuf IPC: the host driver does not support the legacy IPC mode, and RCCL's cross-process
# buffer registration otherwise fails with `hipIpcGetMemHandle: invalid argument` (platform note of the build environment, which exports the
# same value here and on the GPU boxes).  The HSA runtime reads it when it is loaded, so it must be in the environment BEFORE torch
# initialises HIP — setdefault: an exported value wins.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HORIZON = 256
ALGO_BYTES_PER_STEP = 2353      # fp64 state/action in + state/obs/reward/done out, per env-step
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6    # MI355X fp64 vector peak (256 CUs x 4 SIMDs x 16 FMA lanes/clk x 2 x 2.4 GHz)
N_SIMDS = 1024
MAX_CLOCK_HZ = 2.4e9

WORKLOADS = {
    "cfg2": dict(clip="walk", envs=4096, full=False, shards=1, label="BASELINE.json configs[1]: 'walk' mocap, %d envs/GPU, P-controller torque, contacts and limits off"),
    "cfg3": dict(clip="walk", envs=4096, full=True, shards=1, label="BASELINE.json configs[2]: 'walk' mocap, %d envs/GPU, full contact + joint-limit PGS solve, %s reward, RSI auto-reset"),
    "cfg4": dict(clip="spinkick", envs=4096, full=True, shards=8, label="BASELINE.json configs[3]: 'spinkick' mocap, 32768 envs sharded 8 x %d, full contact + joint-limit PGS solve, %s reward, RSI auto-reset, rollout all-gather every 256 steps"),
    "cfg5": dict(clip="dance_b", envs=8192, full=True, shards=8, label="BASELINE.json configs[4]: 'dance_b' mocap, reference-state-init + early termination, 65536 envs sharded 8 x %d, full contact + joint-limit PGS solve, %s reward"),
}


# ---- flop count of the kernel's algorithm ("fp64 work per evaluation") ----------------------------------
# Useful fp64 operations (mul = add = 1, FMA = 2) of ONE forward evaluation as the kernel computes it (tree-sparse factor,
# half-solved rows), as a function of the evaluation's constraint rows `nefc` and PGS sweeps.  Fixed part per evaluation:
#   kinematics 6 600 (28 sincos, hinge chains, 13 frames, cdof, 13 world inertias, composite sums)
#   mass matrix 9 600 (34 I*cdof, 310 entries x 11, 1 432 elimination updates x 3, scaling)
#   bias 4 400 (joint velocity sums, 4-level recursion, 13 body forces, subtree sums, 34 projections)
#   collision 4 200 (16 geom poses, 104 bounding tests, ~20 narrow-phase pairs near the floor)
# and the no-row solve 1 200.  With rows: (nefc + 1) half-solved vectors x 1 232 (Jacobian row 646 + L^-T 552 + scale 34),
# A = Y Y^T 68 nefc^2 (+ b 68 nefc), warm start 2 nefc^2, a PGS sweep 2 nefc^2 + 10 nefc, force assembly 68 nefc, back-solve 620.
FIXED_FLOPS_PER_EVAL = 6600 + 9600 + 4400 + 4200
STEP_OVERHEAD_FLOPS = 2500 + 7000          # RK4 combine / integrate / obs (+ the imitation reward's extra kinematics pass and terms)


def eval_flops(nefc, sweeps):
    nefc = np.asarray(nefc, dtype=np.float64); sweeps = np.asarray(sweeps, dtype=np.float64)
    rows = (nefc + 1) * 1232 + 70 * nefc ** 2 + 136 * nefc + sweeps * (2 * nefc ** 2 + 10 * nefc) + 620
    return FIXED_FLOPS_PER_EVAL + np.where(nefc > 0, rows, 1200.0)


def step_flops(nefc, sweeps):
    """fp64 flops of one env-step: 4 RK evaluations with the row / sweep counts of the last one (what the device reports)."""
    return 4.0 * eval_flops(nefc, sweeps) + STEP_OVERHEAD_FLOPS


def physical_cores():
    """(physical cores, logical CPUs usable by this process) of the host."""
    try:
        avail = sorted(os.sched_getaffinity(0))
    except AttributeError:
        avail = list(range(os.cpu_count() or 1))
    cores = set()
    for c in avail:
        try:
            pk = open("/sys/devices/system/cpu/cpu%d/topology/physical_package_id" % c).read().strip()
            co = open("/sys/devices/system/cpu/cpu%d/topology/core_id" % c).read().strip()
            cores.add((pk, co))
        except OSError:
            cores.add(("?", c))
    return len(cores), len(avail)


def cpu_quota_cores():
    """CPU bandwidth the container may use, in cores (cgroup v2 cpu.max / v1 cfs quota); None = unlimited / unknown."""
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            q, per = open(path).read().split()[:2]
            return None if q == "max" else round(float(q) / float(per), 2)
        except (OSError, ValueError):
            pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else round(q / per, 2)
    except (OSError, ValueError):
        return None


def cpu_baseline_worker(clip, reward, nthreads, seconds):
    """Runs in a fresh process whose OpenMP runtime was configured by the parent (OMP_NUM_THREADS / OMP_PROC_BIND / OMP_PLACES):
    time the CPU oracle (oracle/, float64 C) on a bounded sample of the same workload.  The whole loop — actions, steps, rewards,
    RSI resets on done — runs in C (`dmo_bench_rollout`), whole trajectories per OpenMP thread: round 1's Python-side loop (action
    sampling, per-env resets through ctypes) serialised the run beyond a few dozen threads."""
    from oracle import oracle as O
    from deepmimic_mujoco_amd import MocapDM, CompiledModel, humanoid_spec
    from deepmimic_mujoco_amd.imitation import ImitationSpec
    mc = MocapDM(); mc.load_mocap(clip)
    om = O.Model()
    table = params = None
    if reward == "imitation":
        table, params = ImitationSpec(CompiledModel(humanoid_spec())).table_for(mc)
    n = max(8, nthreads * 4)
    ds = [O.Data(om) for _ in range(n)]
    t0 = time.perf_counter()
    tot, _nd, _rs = O.bench_rollout(om, ds, 40, mc.data_config, mc.data_vel, table, params, 0.9, 1, nthreads)    # calibration (and page-in)
    rate = tot / (time.perf_counter() - t0)
    steps = int(max(40, min(20000, seconds * rate / n)))
    t0 = time.perf_counter()
    tot, nd, rs = O.bench_rollout(om, ds, steps, mc.data_config, mc.data_vel, table, params, 0.9, 2, nthreads)
    el = time.perf_counter() - t0
    print(json.dumps({"rate": tot / el, "n": n, "steps": steps, "el": el, "episodes": nd, "mean_reward": rs / max(1, tot)}))


def cpu_baseline(clip, reward="imitation", budget_s=14.0):
    """The oracle on the box's host cores.  Every thread count runs in its own process with threads pinned
    (OMP_PROC_BIND=close, OMP_PLACES=cores): un-pinned, the runtime piles threads of a 256-logical-CPU host onto a few cores
    (round 1: 129 k env-steps/s at 32 threads, 14 k at 256).  Reported: the physical-core run, next to one core."""
    ncores, nlogical = physical_cores()
    quota = cpu_quota_cores()
    counts = {1, min(32, ncores), ncores, nlogical}
    if quota:                       # the container's CPU-time quota in cores: a run with exactly that many threads (more threads only share the same CPU time)
        counts.add(max(1, min(ncores, int(quota + 0.5))))
    counts = sorted(counts)
    per = budget_s / len(counts)
    res = {}
    for c in counts:
        env = dict(os.environ, OMP_NUM_THREADS=str(c), OMP_PROC_BIND="close", OMP_PLACES="cores" if c <= ncores else "threads", OMP_DYNAMIC="false")
        try:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--_cpu-worker", clip, reward, str(c), str(per)], env=env,
                                 capture_output=True, text=True, timeout=per * 6 + 120)
            res[c] = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception as e:
            res[c] = {"rate": 0.0, "n": 0, "steps": 0, "el": 0.0, "error": repr(e)}
    best = max(res, key=lambda c: res[c]["rate"])
    b = res[best]
    return {"value": round(b["rate"], 1), "unit": "env-steps/s", "cores": best, "kind": "port", "single_core_value": round(res[1]["rate"], 1),
            "by_threads": {str(c): round(res[c]["rate"], 1) for c in counts}, "physical_cores": ncores, "logical_cpus": nlogical,
            "cgroup_cpu_quota_cores": quota,
            "cores_note": ("`cores` is the THREAD count of the fastest run; the container's CPU-time quota is %s cores, so at most that many cores' worth of CPU time "
                           "was available to it (run with exactly that many threads: %s env-steps/s)" % (quota, res.get(max(1, min(ncores, int(quota + 0.5))), {}).get("rate") and
                                                                                                      round(res[max(1, min(ncores, int(quota + 0.5)))]["rate"], 1))) if quota else None,
            "sample": "%d envs x %d steps of the same workload (%s, contacts+limits, %s reward, N(0,0.9^2) actions, RSI reset on done) run entirely "
                      "in C (oracle/dm_oracle.c dmo_bench_rollout, fp64), one trajectory per OpenMP task, %d threads (OMP_PROC_BIND=close, "
                      "OMP_PLACES=cores unless threads > cores; tried %s), %.1f s"
                      % (b["n"], b["steps"], clip, reward, best, counts, b["el"])}


def single_env_gym_loop(device, seconds=2.0):
    """BASELINE.md's C0 shape: the Python loop of src/trpo.py:47-80 on ONE `DPEnv` (host pointers, one launch per step)."""
    import random
    from deepmimic_mujoco_amd import DPEnv
    random.seed(0)
    env = DPEnv(motion="walk", device=device)
    env.seed(0)
    env.reset(); env.reset_model_init()
    rng = np.random.RandomState(0)
    acs = rng.randn(256, 28) * 0.3
    for t in range(32):
        env.step(acs[t])
    steps = eps = 0
    t0 = time.perf_counter()
    while True:
        ob, r, d, _ = env.step(acs[steps % 256])
        steps += 1
        if d:
            env.reset(); env.reset_model_init(); eps += 1
        if steps % 64 == 0 and time.perf_counter() - t0 > seconds:
            break
    el = time.perf_counter() - t0
    env.close()
    return {"value": round(steps / el, 1), "unit": "env-steps/s", "us_per_step": round(el / steps * 1e6, 1), "steps": steps, "episodes": eps,
            "what": "Python `DPEnv.step` loop, 1 env, numpy in / out through the C ABI's host-pointer path (action H2D, one launch, "
                    "one packed obs+reward+done D2H), reset() + reset_model_init() on done as src/trpo.py:78-79"}


def _export_raw(cur, name, outdir, max_rows=4000):
    """DM_PROFILE_KEEP=<dir>: keep a trimmed raw export of a rocprofv3 pass (per-launch kernel rows; per-launch counter values) as CSV,
    so that every figure the line derives from it can be re-derived from the repository (profiles/raw/).  Never raises."""
    import csv
    try:
        os.makedirs(outdir, exist_ok=True)
        cols = [d[0] for d in cur.execute("select * from kernels limit 1").description]
        want = [c for c in ("name", "start", "end", "grid_size", "grid_size_x", "grid_x", "workgroup_size", "workgroup_size_x", "workgroup_x", "vgpr_count",
                            "accum_vgpr_count", "sgpr_count", "lds_size", "scratch_size", "queue_id", "stream_id") if c in cols]
        rows = cur.execute("select %s from kernels order by start limit ?" % ", ".join('"%s"' % c for c in want), (max_rows,)).fetchall()
        with open(os.path.join(outdir, "%s_kernels.csv" % name), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(want + ["duration_ns"])
            i0, i1 = want.index("start"), want.index("end")
            for r in rows:
                r = list(r); r[0] = str(r[0]).split("(")[0][:60]
                w.writerow(r + [r[i1] - r[i0]])
        open(os.path.join(outdir, "%s_kernels.columns.txt" % name), "w").write("all columns of the `kernels` view: " + ", ".join(cols) + "\n")
        if name != "trace":
            rows = cur.execute("select kernel_name, counter_name, value from counters_collection where kernel_name like 'k_%' limit ?", (4 * max_rows,)).fetchall()
            with open(os.path.join(outdir, "%s_counters.csv" % name), "w", newline="") as f:
                w = csv.writer(f)
                w.writerow(["kernel", "counter", "value (one row per launch and counter)"])
                for r in rows:
                    w.writerow([str(r[0]).split("(")[0][:60], r[1], r[2]])
    except Exception as e:
        try:
            open(os.path.join(outdir, "%s_export.err" % name), "w").write(repr(e))
        except Exception:
            pass


def pmc_passes(argv_tail, kernel_prefix, timeout_s=150):
    """Live rocprofv3 passes of THIS workload (short run, child processes): kernel trace, HBM bytes, SQ instruction mix.
    Counters are collected in separate passes with --kernel-trace only (MI355X_MICROARCH.md, rocprofv3 PMC section)."""
    import glob
    import shutil
    import sqlite3
    import tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not on PATH"
    if os.environ.get("ROCP_TOOL_LIBRARIES") or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None, "already running under a profiler: nested counter passes skipped"
    groups = [("trace", ["--stats"]),
              ("fetch", ["--pmc", "FETCH_SIZE"]), ("write", ["--pmc", "WRITE_SIZE"]),
              ("sq", ["--pmc", "SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"]),
              ("grbm", ["--pmc", "GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES"])]
    res = {}
    tmp = tempfile.mkdtemp(prefix="dmenv_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    t0 = time.perf_counter()
    for name, flags in groups:
        if time.perf_counter() - t0 > timeout_s:
            res[name + "_error"] = "time budget spent"
            continue
        d = os.path.join(tmp, name)
        cmd = [exe, "--kernel-trace"] + flags + ["-d", d, "--", sys.executable, os.path.abspath(__file__)] + argv_tail
        try:
            subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=90)
            dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
            if not dbs:
                res[name + "_error"] = "no output database"
                continue
            cur = sqlite3.connect(dbs[0]).cursor()
            keep = os.environ.get("DM_PROFILE_KEEP")
            if keep:
                _export_raw(cur, name, keep)
            if name == "trace":
                r = cur.execute("select count(*), avg(end-start), max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size) "
                                "from kernels where name like ?", (kernel_prefix + "%",)).fetchone()
                res["trace"] = {"launches": r[0], "avg_us": r[1] / 1e3 if r[1] else None, "vgpr": r[2], "agpr": r[3], "sgpr": r[4], "lds_bytes": r[5], "scratch_bytes_per_lane": r[6]}
            else:
                for cn, val, cnt in cur.execute("select counter_name, avg(value), count(*) from counters_collection where kernel_name like ? group by counter_name",
                                                (kernel_prefix + "%",)).fetchall():
                    res[cn] = val
                if name == "sq":    # the same pass's kernel durations: PMC and time from ONE run (a profiled run clocks differently)
                    r = cur.execute("select avg(end-start) from kernels where name like ?", (kernel_prefix + "%",)).fetchone()
                    res["sq_pass_avg_us"] = r[0] / 1e3 if r and r[0] else None
        except Exception as e:
            res[name + "_error"] = repr(e)[:200]
    shutil.rmtree(tmp, ignore_errors=True)
    return res, None


def rollout_bench(args, dev, rank, world, local_dev):
    """Informational: the learner-facing loop of src/trpo.py:27-94 kept on the device — 2x100 tanh policy + value forward,
    Gaussian sampling, env kernel, segment bookkeeping, GAE per 256-step segment.  Not the judged metric line.
    --pipeline P > 1: the rank's envs are P batches whose policy -> env chains run concurrently on their own streams."""
    import torch
    from deepmimic_mujoco_amd import DPVecEnv, MlpPolicy, traj_segment_generator, pipelined_segment_generator, add_vtarg_and_adv
    n = args.envs or 4096
    P = max(1, min(args.pipeline, 8))
    clip = args.clip or "walk"
    pol = MlpPolicy(device=dev, seed=0); pol.seed(rank)
    horizon_form = False
    if not args.unfused:        # one batch, P pipelined sub-batches, the policy step inside the env step kernel: one launch per step
        from deepmimic_mujoco_amd import _abi as A
        env = DPVecEnv(n, motion=clip, device=local_dev, reward="alive", autoreset="init", seed=0, env_offset=rank * n,
                       packed=None if args.packed is None else bool(args.packed))
        env.batch.set_option(A.OPT_PIPELINE, min(P, A.MAX_PIPELINE))
        # the collector steps a fused segment through dm_batch_rollout — one horizon launch where the env leaves the kernel choice open or pins the packed one
        horizon_form = bool(getattr(env, "horizon_packed_ok", False)) or bool(env.packed)
        gen = traj_segment_generator(pol, env, HORIZON, stochastic=True, fused=True)
    elif P > 1:
        cuts = [n * h // P for h in range(P + 1)]
        envs = [DPVecEnv(cuts[h + 1] - cuts[h], motion=clip, device=local_dev, reward="alive", autoreset="init", seed=0, env_offset=rank * n + cuts[h])
                for h in range(P)]
        gen = pipelined_segment_generator(pol, envs, HORIZON, stochastic=True)
    else:
        env = DPVecEnv(n, motion=clip, device=local_dev, reward="alive", autoreset="init", seed=0, env_offset=rank * n)
        gen = traj_segment_generator(pol, env, HORIZON, stochastic=True)
    segs = max(1, args.steps // HORIZON)
    for _ in range(max(3, args.warmup // HORIZON)):            # (the first segments carry one-time costs: allocator growth, lazy kernel loads)
        add_vtarg_and_adv(next(gen), 0.995, 0.97)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eps = 0
    for _ in range(segs):
        seg = add_vtarg_and_adv(next(gen), 0.995, 0.97)
        eps += len(seg["ep_lens"])
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if rank == 0:
        print(json.dumps({"metric": "rollout env-steps/sec (policy in the loop + GAE)", "value": round(world * n * segs * HORIZON / el, 1),
                          "unit": "env-steps/s", "n_gpus": world, "steps": segs * HORIZON, "ms_per_step": round(el / (segs * HORIZON) * 1e3, 4),
                          "episodes": eps, "horizon_launch": (not args.unfused) and horizon_form, "packed_redo_env_steps": env.batch.redo_total() if not args.unfused else None,
                          "config": {"workload": "rollout: %d envs/GPU, %s, untrained 2x100 tanh policy, alive reward, "
                                                                  "noisy-init autoreset, %d-step segments + GAE(0.995, 0.97)"
                                                                  % (n, ("%d concurrently stepped batch(es), one policy launch + one env launch per step" % P) if args.unfused
                                                                     else (("policy step inside the horizon launch: four environments per wavefront, each wave runs its %d steps at its own pace (dm_batch_rollout, k_rollout_packed)" % HORIZON)
                                                                           if horizon_form else ("policy step fused into the env step kernel (one launch per step), %d pipelined sub-batch(es)" % P)), HORIZON)}}))


def standing_bench(args, dev, rank, world, local_dev):
    """`--workload standing`: the regime the reference's own trained policy lives in (src/checkpoint_tmp/DeepMimic/trpo-walk-0, a copy under
    tests/golden/ckpt): the shipped 2x100 tanh policy in the loop (stochastic, as src/trpo.py:49 samples it), noisy-init resets as the trainer's
    episode protocol (src/trpo.py:77-79), 256-step segments.  A population that stands on both feet holds 8 foot corners x 4 pyramid edges = 32
    constraint rows most of the time — the packed kernels' two-row-set path, and beyond 32 rows their in-wave re-step.  Three legs on the same
    protocol: the kernel choice left to the collector (what training gets), four environments per wavefront pinned, one per wavefront pinned."""
    import torch
    from deepmimic_mujoco_amd import DPVecEnv, MlpPolicy, _abi as A
    from deepmimic_mujoco_amd.rollout import SegmentCollector
    n = args.envs or 4096
    clip = args.clip or "walk"
    ckpt = os.path.join(ROOT, "tests", "golden", "ckpt", "trpo-walk-0")
    segs = max(2, args.steps // HORIZON)
    legs = {}
    hist = np.zeros(65, dtype=np.int64)
    for name, packed in (("auto", None), ("packed", True), ("one_env", False)):
        pol = MlpPolicy.from_tf_checkpoint(ckpt, device=dev); pol.seed(rank)
        env = DPVecEnv(n, motion=clip, device=local_dev, reward="alive", autoreset="init", seed=0, env_offset=rank * n, packed=packed)
        env.batch.set_option(A.OPT_PIPELINE, max(1, min(args.pipeline, A.MAX_PIPELINE)))
        col = SegmentCollector(pol, env, HORIZON, True, None, "init", fused=True)

        def next_seg():
            col.launch()
            return col.collect()
        for _ in range(max(3, args.warmup // HORIZON)):          # the population settles into its steady mix of standing / falling / fresh episodes
            next_seg()
        r0 = env.batch.redo_total()
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        t0 = time.perf_counter()
        eps = 0; lens = []
        for _ in range(segs):
            seg = next_seg()
            eps += len(seg["ep_lens"]); lens += seg["ep_lens"]
            if name == "auto":                                    # rows of every env's last evaluation, once per segment (one small D2H)
                hist += np.bincount(np.minimum(env.batch.get(A.F_NEFC), 64), minlength=65)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        redo = env.batch.redo_total() - r0
        legs[name] = {"value": round(world * n * segs * HORIZON / el, 1), "ms_per_step": round(el / (segs * HORIZON) * 1e3, 4), "episodes": eps,
                      "mean_episode_length": round(float(np.mean(lens)), 1) if lens else None, "envs_per_wavefront_at_end": 4 if (col._packed_now if col._packed_now is not None else env.packed) else 1,
                      "env_steps_beyond_packed_capacity": redo, "redo_rate": round(redo / float(n * segs * HORIZON), 6)}
        if name == "auto":
            legs[name]["kernel_switches"] = col.kernel_switches
        env.close()
    if rank == 0:
        tot = max(1, int(hist.sum()))
        out = {"metric": "env-steps/sec, shipped policy in the loop (standing population)", "value": legs["auto"]["value"], "unit": "env-steps/s", "n_gpus": world,
               "steps": segs * HORIZON, "warmup": max(3, args.warmup // HORIZON) * HORIZON, "ms_per_step": legs["auto"]["ms_per_step"], "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "standing: '%s' mocap, %d envs/GPU, full contact + joint-limit PGS solve, alive reward, the reference's shipped trpo-walk-0 policy "
                                      "(stochastic) + value net inside the step kernels, noisy-init reset on done, %d-step segments (dm_batch_rollout)" % (clip, n, HORIZON),
                          "envs_per_gpu": n, "clip": clip},
               "legs": legs,
               "packed_over_one_env": round(legs["packed"]["value"] / legs["one_env"]["value"], 3),
               "rows_per_env_histogram": {"sampled": "nefc of every env's last evaluation at the end of each segment of the `auto` leg", "n": tot,
                                          "frac_le_16": round(float(hist[:17].sum()) / tot, 4), "frac_17_32": round(float(hist[17:33].sum()) / tot, 4),
                                          "frac_gt_32": round(float(hist[33:].sum()) / tot, 4), "mean": round(float((hist * np.arange(65)).sum()) / tot, 2)}}
        print(json.dumps(out))


DISTINCT_GPUS = [1]    # how many physical GPUs the ranks sit on (None: this torch build exposes neither PCI ids nor a device uuid)


def distinct_gpus(rows):
    """rows: per rank [pci domain, bus, device, local hip ordinal, uuid-derived integer].  Distinct PCI addresses when the build reports them; else distinct
    uuids; else distinct local ordinals (one node: the ordinal IS the GPU); None when nothing tells the ranks' devices apart."""
    if all(min(r[:3]) >= 0 for r in rows):
        return len(set(tuple(r[:3]) for r in rows))
    if all(r[4] != 0 for r in rows):
        return len(set(r[4] for r in rows))
    return None


RANK_DEVICES = []      # per rank: PCI address of its GPU, gathered over the process group at start-up (multi-rank runs)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--_cpu-worker":
        return cpu_baseline_worker(sys.argv[2], sys.argv[3], int(sys.argv[4]), float(sys.argv[5]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=512)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--envs", type=int, default=0, help="envs per GPU (default: the workload's: 4096; cfg5 8192)")
    ap.add_argument("--workload", default="cfg3", choices=["cfg2", "cfg3", "cfg4", "cfg5", "rollout", "standing"])
    ap.add_argument("--clip", default=None, help="override the workload's mocap clip")
    ap.add_argument("--reward", default="imitation", choices=["imitation", "v3-config", "alive"],
                    help="reward of the full-contact workloads: the 5-term DeepMimic imitation reward (default), dp_env_v3's config reward, or the constant 1.0")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (one GPU per rank); gloo = host-staged gather, ranks may share a GPU (LOCAL_RANK modulo the visible devices)")
    ap.add_argument("--pipeline", type=int, default=2,
                    help="DM_OPT_PIPELINE: sub-batches per GPU stepped on their own streams so that one's drain overlaps the next one's ramp "
                         "across consecutive steps (1 = one launch per step)")
    ap.add_argument("--unfused", action="store_true", help="rollout workload: separate policy launch per step (the round-1 form) instead of the fused step")
    ap.add_argument("--dtype", type=int, default=64, choices=[64, 32],
                    help="arithmetic of the kernels: 64 (the parity path, the judged line) or 32 (libdmenv32.so, the float32 build: informational)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the live rocprofv3 counter passes (N = 1 only; they add about a minute)")
    ap.add_argument("--no-gym-loop", action="store_true", help="skip the single-env Python DPEnv.step loop (N = 1 only; ~3 s)")
    ap.add_argument("--prewarm-horizons", type=int, default=6, help="untimed 256-step horizons before the warm-up steps (cold-box clock ramp, ~1 s)")
    ap.add_argument("--horizon-launch", action="store_true",
                    help="step through dm_batch_rollout: up to one %d-step horizon of pre-drawn actions per call (on the packed path ONE launch in which every "
                         "wavefront runs its four environments through all steps at its own pace) instead of one dm_batch_step call per step; implies --packed 1 unless given" % HORIZON)
    ap.add_argument("--step-queue", type=int, default=HORIZON,
                    help="DM_OPT_STEP_QUEUE of the timed dm_batch_step calls: up to Q calls are queued and run as ONE horizon launch (every wavefront steps its "
                         "four environments through the queued steps at its own pace) when the queue is full or the caller joins; implies four environments per "
                         "wavefront.  0 = every call launches (rounds 1-3).  Applies to the full-contact workloads at up to 8192 envs per GPU")
    ap.add_argument("--repeats", type=int, default=3, help="timed windows of --steps steps each; `value` is the median window, `value_spread` min / median / max")
    ap.add_argument("--no-vecenv-leg", action="store_true", help="skip the leg that times the same steps through DPVecEnv.step, one launch per call")
    ap.add_argument("--no-horizon-leg", action="store_true", help="skip the second timed leg (the same steps through dm_batch_rollout, reported as `horizon_launch`)")
    ap.add_argument("--rollout-form", default="auto", choices=["auto", "launch", "steps"],
                    help="dm_batch_rollout on the packed path (DM option 106): one launch per call, the library's step launches, or chosen by batch size (default)")
    ap.add_argument("--horizon-chunk", type=int, default=HORIZON, help="--horizon-launch: steps per dm_batch_rollout call (the dispatch order — which environments share a wavefront — is renewed between calls)")
    ap.add_argument("--packed", type=int, default=None, choices=[0, 1], help="DM option 105: four environments per wavefront (k_step_packed) where that kernel covers the workload (default: the library's)")
    ap.add_argument("--no-reorder", action="store_true", help="experiment: identity dispatch order instead of longest-first (DM option 104 = 0)")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the multi-rank code path with however many ranks there are, ONE included: process group (RCCL with --dist-backend nccl), "
                         "double-buffered device all-gather of the rollout block every 256 steps, max-reduced timing, the learner's all-mean on a gradient-sized vector")
    ap.add_argument("--alloc-gather-world", type=int, default=0, help="with --force-dist: also allocate (and touch) the two G-way gathered rollout buffers a G-rank job holds per rank")
    ap.add_argument("--_child", action="store_true", help=argparse.SUPPRESS)   # profiled child of pmc_passes: GPU loop only, prints nothing
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from deepmimic_mujoco_amd import DPVecEnv, _abi as A

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE is %d: launch N > 1 as `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    ndev = torch.cuda.device_count()
    if args.dist_backend == "nccl" and world > ndev:
        raise SystemExit("RCCL needs one GPU per rank: %d ranks, %d devices visible (use --dist-backend gloo to share a GPU)" % (world, ndev))
    local_dev = local_rank % ndev
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    n_ranks_seen = 1
    dist_on = world > 1 or args.force_dist
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        n_ranks_seen = dist.get_world_size()
        assert n_ranks_seen == args.gpus, "process group has %d ranks, --gpus says %d" % (n_ranks_seen, args.gpus)
        # which physical GPU every rank sits on (PCI domain:bus:device of its HIP device), gathered over the process group itself: a SCALE record then
        # shows that the collective backend saw N distinct GPUs (src/train_mpi.sh:1 starts one worker per slot; src/trpo.py:175-186 sums over them)
        pr = torch.cuda.get_device_properties(dev)
        pci = [int(getattr(pr, k, -1)) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")]
        # a torch build without the PCI properties: the device's uuid (its first 8 bytes as an integer) keys distinctness instead, and failing that the
        # count is reported as unknown (None) — never as "1 GPU" for N real ones
        uid = 0
        try:
            uid = int.from_bytes(bytes.fromhex(str(pr.uuid).replace("-", "").replace("GPU", ""))[:7], "big")
        except Exception:       # noqa: BLE001
            uid = 0
        mine = torch.tensor(pci + [local_dev, uid], dtype=torch.int64, device=dev if args.dist_backend == "nccl" else "cpu")
        every = [torch.zeros_like(mine) for _ in range(n_ranks_seen)]
        dist.all_gather(every, mine)
        RANK_DEVICES[:] = ["%04x:%02x:%02x.hip%d" % tuple(int(v) & 0xffff for v in t.tolist()[:4]) for t in every]
        DISTINCT_GPUS[0] = distinct_gpus([t.tolist() for t in every])
        if rank == 0:       # start-up line for the scaling log (stderr: stdout carries exactly one JSON line)
            sys.stderr.write("bench.py: rank -> GPU (pci domain:bus:device.hip-ordinal): %s; %s distinct GPU(s) for %d rank(s)\n"
                             % (" ".join("%d=%s" % (i, d) for i, d in enumerate(RANK_DEVICES)), "unknown number of" if DISTINCT_GPUS[0] is None else DISTINCT_GPUS[0], n_ranks_seen))
            n_log = args.envs or WORKLOADS.get(args.workload, WORKLOADS["cfg3"])["envs"]
            sys.stderr.write("bench.py: %s process group up, %d ranks (backend reports %s), device %s; rollout gather every %d steps: "
                             "[%d, %d, 87] f32 = %.1f MB per rank, %.1f MB gathered per rank\n"
                             % (args.dist_backend, n_ranks_seen, dist.get_backend(), dev, HORIZON, HORIZON, n_log, HORIZON * n_log * 87 * 4 / 1e6,
                                world * HORIZON * n_log * 87 * 4 / 1e6))
            sys.stderr.flush()

    if args.workload == "rollout":
        return rollout_bench(args, dev, rank, world, local_dev)
    if args.workload == "standing":
        return standing_bench(args, dev, rank, world, local_dev)
    wl = WORKLOADS[args.workload]
    n = args.envs or wl["envs"]
    clip = args.clip or wl["clip"]
    full = wl["full"]
    # shard id: the rank when the job is sharded; a single process of a sharded configuration times an interior shard (3), so
    # that the measured shard is not the one whose global env ids start at 0
    shard = rank if world > 1 else (3 if wl["shards"] > 1 else 0)
    # the timed dm_batch_step calls are queued (DM_OPT_STEP_QUEUE) where the library runs a queue as one horizon launch: the packed kernels, a model
    # with constraint rows, at most two packed waves per SIMD
    queue = max(0, min(args.step_queue, A.MAX_STEP_QUEUE)) if (full and args.dtype == 64 and n <= 8192 and args.packed != 0 and not args.horizon_launch) else 0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", UserWarning)        # frame_skip = 1 with the imitation reward is this benchmark's definition of a step
        env = DPVecEnv(n, motion=clip, device=local_dev, reward=args.reward if full else "alive",
                       autoreset="rsi", seed=0, contacts=full, limits=full,
                       action_mode="raw" if full else "p-control", env_offset=shard * n, frame_skip=1, dtype=args.dtype,
                       packed=(True if (args.horizon_launch or queue) else None) if args.packed is None else bool(args.packed))
    from deepmimic_mujoco_amd.dp_env import PACKED_FROM_ENVS
    default_packed = (n >= PACKED_FROM_ENVS or not full) and args.dtype == 64          # what DPVecEnv(packed=None) starts on
    step_kernel = ("k_rollout_packed" if ((args.horizon_launch or queue) and full and n <= 8192) else "k_step_packed") if env.packed else "k_step_narrow"
    stream = torch.cuda.Stream(device=dev)
    env.batch.set_stream(stream.cuda_stream)
    env.batch.set_option(A.OPT_PIPELINE, max(1, min(args.pipeline, A.MAX_PIPELINE)))
    if queue:
        env.batch.set_option(A.OPT_STEP_QUEUE, queue)
    if args.no_reorder:
        env.batch.set_option(104, 0)
    if args.rollout_form != "auto":
        env.batch.set_option(106, 1 if args.rollout_form == "launch" else 0)

    with torch.cuda.stream(stream):
        gen = torch.Generator(device=dev); gen.manual_seed(1234 + shard)
        pool = HORIZON       # one 256-step horizon of i.i.d. N(0, 0.9^2) actions per (env, t), reused from horizon to horizon
        if full:
            actions = torch.randn((pool + 1, n, A.NU), generator=gen, device=dev, dtype=torch.float64) * 0.9     # (row `pool`: never read, dm_batch_rollout's T + 1 shape)
        else:
            actions = torch.zeros((pool + 1, n, A.NU), device=dev, dtype=torch.float64)   # cfg2: pure P-controller
        # the kernel writes obs / reward / done of step t straight into row t of [T, n, .] staging buffers (no per-step copy
        # kernels); at the end of each 256-step horizon they are packed into the f32 rollout block (obs 56 + act 28 + rew + done
        # + vpred) in one go.  Blocks are double-buffered: while one is all-gathered (async, on the collective's own stream) the
        # envs keep stepping
        from deepmimic_mujoco_amd.rollout import DoubleBufferedGather
        obs_T = torch.zeros((HORIZON, n, A.NOBS), dtype=torch.float64, device=dev)    # zeros: every page is touched before the clock starts
        rew_T = torch.zeros((HORIZON, n), dtype=torch.float64, device=dev)
        done_T = torch.zeros((HORIZON, n), dtype=torch.uint8, device=dev)
        tidx = torch.arange(HORIZON, device=dev)
        dbg = DoubleBufferedGather(HORIZON, n, device=dev, world=world, collective=dist_on)
        gather_alloc = None
        if dist_on and args.alloc_gather_world > 1:                # what a G-rank job holds per rank beside the two blocks: allocated and touched
            gather_alloc = [torch.zeros((args.alloc_gather_world * HORIZON, n, 87), dtype=torch.float32, device=dev) for _ in range(2)]
        env.reset("rsi")
        stat_nefc, stat_iter = [], []

        act_rows = [actions[k] for k in range(pool)]                       # the per-step buffers, as any caller holds them: views made once
        out_rows = [(obs_T[k], rew_T[k], done_T[k]) for k in range(HORIZON)]

        def one_step(t, stats=False):
            k = t % HORIZON
            env.batch.step(act_rows[t % pool], 1, out_rows[k])
            if k == HORIZON - 1:
                env.batch.join()                                              # pipelined sub-batches: this stream now consumes their outputs
                blk = dbg.block(t)
                blk[:, :, :56] = obs_T; blk[:, :, 56:84] = actions[(tidx + (t - k)) % pool]
                blk[:, :, 84] = rew_T; blk[:, :, 85] = done_T
                dbg.commit(t)

        drain = dbg.drain

        def run_steps(t_from, t_to, horizon=None):
            """steps t_from .. t_to - 1: one dm_batch_step call each, or (--horizon-launch) one dm_batch_rollout call per stretch inside a horizon"""
            if not (args.horizon_launch if horizon is None else horizon):
                for t in range(t_from, t_to):
                    one_step(t)
                return
            t = t_from
            while t < t_to:
                k = t % HORIZON
                m = min(HORIZON - k, t_to - t, max(1, args.horizon_chunk))
                env.batch.rollout(actions[k:k + m + 1], (obs_T[k:k + m], rew_T[k:k + m], done_T[k:k + m]), 1)
                t += m
                if t % HORIZON == 0:
                    env.batch.join()
                    blk = dbg.block(t - 1)
                    blk[:, :, :56] = obs_T; blk[:, :, 56:84] = actions[:pool]
                    blk[:, :, 84] = rew_T; blk[:, :, 85] = done_T
                    dbg.commit(t - 1)

        # untimed: bring a cold box (first process after boot: idle clocks, unmapped VRAM) to its steady state, then the W warm-up steps
        # (a fixed number of horizons, so that every rank issues the same sequence of collectives)
        for _ in range(args.prewarm_horizons):
            run_steps(0, HORIZON)
            drain(); env.batch.sync()
        run_steps(0, args.warmup)
        drain()
        env.batch.sync()

        issue_s = []

        def timed_window(fn):
            """one timed window: barrier + synchronize on both sides; returns (host seconds, HIP-event ms on the launch stream)"""
            if dist_on:
                dist.barrier()
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            w0 = time.perf_counter()
            e0.record(stream)
            fn()
            issue_s.append(time.perf_counter() - w0)                       # host time to issue the window's calls (before the join that ends it)
            env.batch.join()                                               # queued steps run, every sub-batch launch is inside the events
            drain()                                                        # outstanding gathers finish inside the timed region
            e1.record(stream)
            stream.synchronize()
            torch.cuda.synchronize()
            if dist_on:
                dist.barrier()
            return time.perf_counter() - w0, e0.elapsed_time(e1)

        reps = 1 if args._child else max(1, args.repeats)
        windows = [timed_window(lambda: run_steps(0, args.steps)) for _ in range(reps)]
        host_issue_us = float(np.median(issue_s[:reps])) / args.steps * 1e6
        # per-launch duration of the step kernel by HIP events on the stream it is launched on (pipelined: sub-batch 0's launch on
        # its own stream, while the other sub-batches keep the machine busy): untimed, sampled after the clock stopped
        launch_us = []
        queue_launch_steps = min(queue, args.steps) if queue else 0        # steps one horizon launch of the timed window carries
        if not args._child and queue:
            env.batch.enable_timing(True)
            for _ in range(3):                                             # the horizon launch of a window-sized queue, by the library's events around it
                run_steps(0, queue_launch_steps); env.batch.join()
                launch_us.append(env.batch.last_step_ms() * 1e3)
            env.batch.enable_timing(False)
            drain(); env.batch.sync()
        elif not args._child and not args.horizon_launch:
            env.batch.enable_timing(True)
            for t in range(args.steps, args.steps + 48):
                one_step(t)
                if t >= args.steps + 8:
                    launch_us.append(env.batch.last_step_ms() * 1e3)
            env.batch.enable_timing(False)
            env.batch.sync()
        # row / sweep statistics for the flop count: untimed, a few more steps sampled after the clock stopped
        if not args._child:
            for t in range(args.steps, args.steps + 8):
                one_step(t)
                env.batch.sync()
                stat_nefc.append(env.batch.get(A.F_NEFC)); stat_iter.append(env.batch.get(A.F_SOLVER_ITER))
            drain()
        # ---- further legs, reported beside the judged value (same state stream, same pre-drawn actions; max over ranks, same barriers) ----------
        # (1) `vecenv_step`: the same number of steps through the facade `DPVecEnv.step` (the drop-in for VecEnv.step, src/utils/vec_env/__init__.py:
        #     26-100), nothing queued: every call launches, on the kernel DPVecEnv(packed=None) picks for this batch size
        ve_elapsed = None; ve_kernel = None
        if not args._child and args.dtype == 64 and not args.horizon_launch and not args.no_vecenv_leg:
            bt = env.batch
            was_packed = env.packed
            if queue:
                bt.set_option(A.OPT_STEP_QUEUE, 0)
            if args.packed is None:
                bt.set_option(A.OPT_PACKED, 1 if default_packed else 0)
            ve_kernel = "k_step_packed" if env.packed else "k_step_narrow"

            def facade_steps():
                for t in range(args.steps):
                    k = t % HORIZON
                    env.step(act_rows[t % pool], out=out_rows[k])
                    if k == HORIZON - 1:
                        bt.join()
            facade_steps(); bt.join(); bt.sync()                               # untimed: first use of this kernel
            ve_elapsed = sorted(timed_window(facade_steps)[0] for _ in range(reps))[reps // 2]
            bt.set_option(A.OPT_PACKED, 1 if was_packed else 0)
            if queue:
                bt.set_option(A.OPT_STEP_QUEUE, queue)
        # (2) `horizon_launch`: whole 256-step horizons through dm_batch_rollout — one launch per horizon in which every wavefront runs its four
        #     environments through all steps at its own pace (what a queue of 256 steps also becomes)
        hl_elapsed = None; hl_redo = None
        hl_steps = (args.steps + HORIZON - 1) // HORIZON * HORIZON          # whole horizons: a horizon launch shorter than a horizon has not averaged its waves yet
        if not args._child and args.dtype == 64 and not args.horizon_launch and not args.no_horizon_leg:
            bt = env.batch
            was_packed = env.packed; was_auto = bool(bt.__dict__.get("_auto", False))
            if was_auto:
                bt.enable_auto_packed(False)
            bt.set_option(A.OPT_PACKED, 1)
            r0 = bt.redo_total()
            run_steps(0, HORIZON, horizon=True); drain(); bt.sync()            # untimed: first use of the kernel
            hl_elapsed = timed_window(lambda: run_steps(0, hl_steps, horizon=True))[0]
            hl_redo = bt.redo_total() - r0
            bt.set_option(A.OPT_PACKED, 1 if was_packed else 0)
            if was_auto:
                bt.enable_auto_packed(True)
    order = sorted(range(reps), key=lambda i: windows[i][0])
    mid = order[reps // 2]                                                  # the median window is the one reported
    elapsed_all = [w[0] for w in windows]
    allmean_ok = None
    if dist_on:
        from deepmimic_mujoco_amd.trpo import allmean
        gvec = torch.full((18656,), float(rank + 1), dtype=torch.float32, device=dev if args.dist_backend == "nccl" else "cpu")   # the flat gradient's size (src/trpo.py:175-180)
        allmean(gvec, force=True)
        allmean_ok = bool(abs(float(gvec[0]) - (world + 1) / 2.0) < 1e-6 and abs(float(gvec[-1]) - (world + 1) / 2.0) < 1e-6)
    if dist_on:
        tt = torch.tensor(elapsed_all + [hl_elapsed or 0.0, ve_elapsed or 0.0], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed_all = [float(x) for x in tt[:reps].tolist()]
        order = sorted(range(reps), key=lambda i: elapsed_all[i]); mid = order[reps // 2]
        if hl_elapsed is not None:
            hl_elapsed = float(tt[reps].item())
        if ve_elapsed is not None:
            ve_elapsed = float(tt[reps + 1].item())
    if args._child:
        return
    elapsed = elapsed_all[mid]
    gpu_ms = windows[mid][1]
    status = env.batch.get(A.F_STATUS)
    nefc = np.concatenate(stat_nefc); iters = np.concatenate(stat_iter)

    if rank == 0:
        total_steps = world * n * args.steps
        value = total_steps / elapsed
        P_sub = max(1, min(args.pipeline, A.MAX_PIPELINE))
        kernel_ms = gpu_ms / args.steps       # per-launch duration on the launch stream (incl. k_order and the per-horizon block packing)
        ach_gbs = ALGO_BYTES_PER_STEP * n / (kernel_ms * 1e-3) / 1e9
        rew_name = "5-term DeepMimic imitation" if args.reward == "imitation" else args.reward
        label = wl["label"] % ((n, rew_name) if full else (n,))
        if wl["shards"] > 1 and world == 1:
            label += "; ONE shard timed (shard %d of %d: global env ids %d..%d, no gather)" % (shard, wl["shards"], shard * n, shard * n + n - 1)
        flops = float(step_flops(nefc, iters).mean())
        tflops = flops * n / (kernel_ms * 1e-3) / 1e12
        out = {
            "metric": "env-steps/sec", "value": round(value, 1), "unit": "env-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f%d" % args.dtype, "data": "synthetic",
            "config": {"workload": label, "envs_per_gpu": n, "global_envs": world * n if world > 1 else n * wl["shards"], "clip": clip,
                       "parallelism": "env-shard x%d" % max(world, wl["shards"]), "n_ranks_seen": n_ranks_seen,
                       "rank_devices": list(RANK_DEVICES) or None, "distinct_gpus": DISTINCT_GPUS[0],
                       "non_default_options": (["DM_OPT_STEP_QUEUE=%d" % queue] if queue else []) + (["DM_OPT_PIPELINE=%d" % P_sub] if (P_sub > 1 and not queue) else []),
                       "value_is": ("queued dm_batch_step calls (open loop, outputs valid after dm_batch_join): the library's horizon launch reached through the per-step entry point; "
                                    "the closed-loop figure — one launch set per call, default options apart from the pipeline depth DPVecEnv sets — is `vecenv_step`") if queue
                                   else "one launch set per dm_batch_step call",
                       "dist_backend": args.dist_backend if dist_on else None, "forced_dist": bool(args.force_dist), "learner_allmean_ok": allmean_ok,
                       "gather_buffers_allocated_bytes": (sum(int(x.numel()) * 4 for x in gather_alloc) if gather_alloc else None),
                       "rollout_allgather_every": HORIZON if dist_on else None, "gathers_completed": dbg.completed,
                       "pipeline_sub_batches": max(1, min(args.pipeline, A.MAX_PIPELINE)),
                       "host_issue_us_per_step": round(host_issue_us, 2),
                       "step_queue": queue, "step_queue_launches": (env.batch.queue_stats()[0] if queue else None),
                       "step_queue_note": ("DM_OPT_STEP_QUEUE = %d: the timed dm_batch_step calls are queued and run as one horizon launch (k_rollout_packed) per %d steps or at the "
                                           "join that ends the window — open-loop stepping with pre-drawn actions, the same contract as DM_OPT_PIPELINE (outputs valid after "
                                           "dm_batch_join); bit-identical to unqueued packed steps (tests/test_gpu_queue.py).  A closed loop that joins every step gets one launch "
                                           "per call: `vecenv_step`" % (queue, queue)) if queue else None,
                       "sim_steps_per_env_step": 1,
                       "mean_nefc": round(float(nefc.mean()), 2), "mean_pgs_sweeps": round(float(iters.mean()), 2),
                       "overflow_envs": int((status & 1).sum()),
                       "envs_per_wavefront": 4 if env.packed else 1, "kernel_switches_by_row_statistics": getattr(env.batch, "auto_switches", None), "packed_redo_env_steps": env.batch.redo_total() if env.packed else None,
                       "actions": ("i.i.d. N(0, 0.9^2) per (env, step) within a %d-step horizon, pre-drawn on the device (%d x %d x 28 f64), the same "
                                   "tensors reused by every horizon" % (pool, pool, n)) if full else "zeros (pure P-controller)",
                       "timed_window": "%s (state / obs / reward / done device-resident)%s" % (
                           ("%d steps through dm_batch_rollout, %d steps per call" % (args.steps, min(HORIZON, max(1, args.horizon_chunk)))) if args.horizon_launch else
                           ("%d dm_batch_step calls (Batch.step), queued: %d horizon launch(es)" % (args.steps, (args.steps + queue - 1) // queue)) if queue else "%d dm_batch_step calls (Batch.step), one launch set per call" % args.steps,
                           ", %d horizon-end block packings + joins" % (args.steps // HORIZON) if args.steps >= HORIZON
                           else "; shorter than the %d-step horizon: no block packing / join inside the window" % HORIZON)},
            "value_spread": {"windows": reps, "steps_per_window": args.steps, "min": round(total_steps / max(elapsed_all), 1), "median": round(value, 1),
                             "max": round(total_steps / min(elapsed_all), 1), "unit": "env-steps/s",
                             "what": "%d timed windows of --steps steps each, every one bracketed by barrier + synchronize (max over ranks); `value` / `ms_per_step` are the median window" % reps},
            "vecenv_step": None if ve_elapsed is None else {
                "value": round(total_steps / ve_elapsed, 1), "unit": "env-steps/s", "ms_per_step": round(ve_elapsed / args.steps * 1e3, 4), "steps": args.steps,
                "kernel": ve_kernel, "envs_per_wavefront": 4 if ve_kernel == "k_step_packed" else 1,
                "what": "the same number of steps of the same state stream through the facade `DPVecEnv.step(actions, out=...)` (deepmimic_mujoco_amd/dp_env.py: step_async + step_wait, the "
                        "drop-in for VecEnv.step) with nothing queued: every call launches at once, on the kernel DPVecEnv(packed=None) picks for this batch size, %d pipelined "
                        "sub-batch(es); median of %d windows.  This is what a caller gets that consumes every step's outputs before the next call" % (P_sub, reps)},
            "horizon_launch": None if hl_elapsed is None else {
                "value": round(world * n * hl_steps / hl_elapsed, 1), "unit": "env-steps/s", "ms_per_step": round(hl_elapsed / hl_steps * 1e3, 4), "steps": hl_steps,
                "steps_per_call": min(HORIZON, max(1, args.horizon_chunk)),
                "kernel": "k_rollout_packed" if (full and n <= 8192) else "k_step_packed: the library issues the step launches itself (no constraint rows, or more than two packed waves per SIMD: one launch per horizon does not pay there)",
                "envs_per_wavefront": 4,
                "env_steps_re_stepped_in_wave": hl_redo,
                "what": "%d steps (the timed window rounded up to whole 256-step horizons) of the same workload and state stream through dm_batch_rollout (max over ranks, same barriers): ONE launch per horizon of pre-drawn actions, every wavefront "
                        "steps its four environments through the whole horizon at its own pace; results bit-identical to the dm_batch_step calls of `value` on the packed "
                        "kernel (tests/test_gpu_rollout.py::test_horizon_launch_equals_step_by_step) and oracle-checked at full shard size "
                        "(tests/test_gpu_fullsize.py [*-2]).  With a step queue (`config.step_queue`) `value` reaches this same kernel through dm_batch_step calls; the "
                        "one-launch-set-per-call figure, the drop-in for VecEnv.step, is `vecenv_step`" % hl_steps},
            "roofline": {"bound": "hbm", "achieved": round(ach_gbs, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach_gbs / HBM_PEAK_GBS, 6), "traffic": None,
                         "kernel": step_kernel, "kernel_ms": round(kernel_ms, 4),
                         "kernel_ms_covers": "HIP events around the timed region / steps: one dm_batch_step = the step kernel (all sub-batches; the packed kernel is followed by k_step_redo; the kernels order themselves — no k_order launch since round 5) + 1/256 of a horizon's block packing; "
                                             "with --pipeline > 1 consecutive steps overlap, so this is the per-step issue interval, not a lone launch's latency",
                         "algorithmic_bytes_per_env_step": ALGO_BYTES_PER_STEP,
                         "launch": ({"kernel": step_kernel, "envs_per_launch": n, "steps_per_launch": min(HORIZON, max(1, args.horizon_chunk), args.steps),
                                     "algorithmic_bytes": ALGO_BYTES_PER_STEP * n * min(HORIZON, max(1, args.horizon_chunk), args.steps),
                                     "avg_us": round(kernel_ms * 1e3 * min(HORIZON, max(1, args.horizon_chunk), args.steps), 1),
                                     "measured": "HIP events around the timed region / launches (one launch = one horizon of all envs; launches do not overlap)"}
                                    if args.horizon_launch else
                                    {"kernel": step_kernel, "envs_per_launch": n, "steps_per_launch": queue_launch_steps,
                                     "algorithmic_bytes": ALGO_BYTES_PER_STEP * n * queue_launch_steps,
                                     "avg_us": round(float(np.mean(launch_us)), 1) if launch_us else None,
                                     "measured": "HIP events the library records around the horizon launch on its stream (dm_batch_last_step_ms), %d launches of %d queued steps "
                                                 "sampled after the timed region" % (len(launch_us), queue_launch_steps)}
                                    if queue else None) or {"kernel": step_kernel, "envs_per_launch": n // P_sub, "launches_per_step": P_sub,
                                    "algorithmic_bytes": ALGO_BYTES_PER_STEP * (n // P_sub),
                                    "avg_us": round(float(np.mean(launch_us)), 1) if launch_us else None,
                                    "measured": "HIP events on the launch's own stream, %d launches sampled after the timed region" % len(launch_us),
                                    "launches_in_flight": round(float(np.mean(launch_us)) * 1e-3 * P_sub / kernel_ms, 2) if launch_us else None,
                                    "note": "`achieved` is the whole-GPU rate, bytes of one step / step interval: with pipelined sub-batches "
                                            "launches overlap, so bytes / one launch's duration is only that launch's share of the machine"},
                         "note": "latency / fp64-issue bound path, not an HBM stream: see the fp64 and VALU fields",
                         "fp64_flops_per_env_step": round(flops, 0),
                         "fp64_flops_source": "flop count of the kernel's algorithm (bench.py eval_flops) on this run's nefc / PGS-sweep samples (%d env-steps)" % nefc.size,
                         "fp64_tflops": round(tflops, 4), "fp64_valu_peak_tflops": FP64_VALU_PEAK_TFLOPS,
                         "fp64_frac": round(tflops / FP64_VALU_PEAK_TFLOPS, 5)},
        }
        if world == 1 and not args.no_pmc:
            hz = args.horizon_launch or bool(queue)          # the profiled child runs exactly one HORIZON-step launch of k_rollout_packed
            tail = ["--workload", args.workload, "--reward", args.reward] + (["--steps", str(HORIZON), "--warmup", "0", "--prewarm-horizons", "0"] + (["--horizon-launch"] if args.horizon_launch else ["--step-queue", str(HORIZON)]) if hz
                                                                              else ["--steps", "48", "--warmup", "8", "--prewarm-horizons", "1", "--step-queue", "0"]) + [
                    "--envs", str(n), "--_child", "--no-pmc", "--no-cpu-baseline", "--no-gym-loop", "--pipeline", str(args.pipeline), "--dtype", str(args.dtype)] + (["--clip", args.clip] if args.clip else [])
            pmc, err = pmc_passes(tail + (["--packed", "1" if env.packed else "0"]), step_kernel)
            r = out["roofline"]
            # counters are averages per LAUNCH: one step is P_sub launches of n / P_sub envs (one call per step), or 1 / HORIZON of a launch of all n envs
            # (--horizon-launch: the profiled child runs exactly one HORIZON-step launch)
            launches_per_step = (1.0 / HORIZON) if hz else float(P_sub)
            envsteps_per_launch = float(n * HORIZON) if hz else float(n // P_sub)
            if pmc is None:
                r["pmc"] = err
            else:
                r["pmc"] = {k: (round(v, 1) if isinstance(v, float) else v) for k, v in pmc.items()}
                if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
                    # FETCH_SIZE / WRITE_SIZE are KiB.  The guide's x2 gfx950 correction of FETCH_SIZE was calibrated on 16 B/lane
                    # streaming reads; this kernel reads 8 B/lane rows, an uncalibrated width: both readings are reported
                    r["traffic"] = round((pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0 * launches_per_step)
                    r["traffic_fetch_doubled"] = round((2 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0 * launches_per_step)
                    r["traffic_over_algorithmic"] = round(r["traffic"] / (ALGO_BYTES_PER_STEP * n), 3)
                    r["traffic_source"] = "live rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload, HBM bytes per step (= %.4g launches)" % launches_per_step
                    if full and args.reward == "imitation":
                        r["traffic_note"] = ("includes the parked kinematics of the imitation modes: 694 doubles written by a step's reward pass and read back by the "
                                             "next step instead of recomputing one kinematics pass in five (5 552 B each way per env-step = %.1f MB per step; +2 %% "
                                             "env-steps/s for 0.6 %% of the HBM peak)" % (2 * 694 * 8 * n / 1e6))
                if pmc.get("SQ_ACTIVE_INST_VALU"):
                    # SQ_ACTIVE_INST_VALU counts quad-cycles summed over all SIMDs; the denominator is the un-profiled step interval
                    r["valu_issue_frac"] = round(launches_per_step * pmc["SQ_ACTIVE_INST_VALU"] * 4.0 / (N_SIMDS * kernel_ms * 1e-3 * MAX_CLOCK_HZ), 4)
                    r["valu_issue_frac_note"] = "launches/step x SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x step interval x 2.4 GHz)"
                if pmc.get("GRBM_GUI_ACTIVE") and pmc.get("trace", {}).get("avg_us"):
                    r["effective_clock_ghz"] = round(pmc["GRBM_GUI_ACTIVE"] / 8.0 / (pmc["trace"]["avg_us"] * 1e-6) / 1e9, 3)   # summed over the 8 XCDs
                if pmc.get("SQ_INSTS_VALU"):
                    per_env = pmc["SQ_INSTS_VALU"] / envsteps_per_launch
                    r["valu_wave_instr_per_env_step"] = round(per_env, 1)
                    # useful fp64 lane-operations (an FMA lane does 2 flops) over the lane slots of all VALU instructions issued
                    r["lane_efficiency"] = round((flops / 2.0) / (per_env * 64.0), 4)
                    r["lane_efficiency_note"] = "useful fp64 FMA-lane operations / (VALU wave-instructions x 64 lanes)"
                if pmc.get("SQ_WAVE_CYCLES") and pmc.get("SQ_WAIT_ANY"):
                    r["wave_wait_frac"] = round(pmc["SQ_WAIT_ANY"] / pmc["SQ_WAVE_CYCLES"], 4)
                    r["wave_valu_frac"] = round(pmc.get("SQ_ACTIVE_INST_VALU", 0.0) / pmc["SQ_WAVE_CYCLES"], 4)
        if world == 1 and not args.no_gym_loop:
            try:
                env.close()
                out["single_env_gym_loop"] = single_env_gym_loop(local_dev)
            except Exception as e:
                out["single_env_gym_loop"] = {"value": None, "error": repr(e)[:200]}
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(clip, args.reward if full else "alive")
            except Exception as e:  # the baseline must never take the GPU number down with it
                out["cpu_baseline"] = {"value": None, "unit": "env-steps/s", "cores": os.cpu_count(), "kind": "port", "sample": "failed: %r" % (e,)}
        print(json.dumps(out))
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
