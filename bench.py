#!/usr/bin/env python3
"""bench.py — env-steps/s of the batched DeepMimic humanoid step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

Workload (config.workload): BASELINE.json configs[2] — 'walk' clip, 4096 envs per GPU, full contact + joint-limit
solve (PGS 50), the 5-term DeepMimic imitation reward (pose / velocity / end-effector / root / COM against the mocap frame;
`--reward v3-config` selects dp_env_v3's own disabled config reward instead), RSI auto-reset on done, actions ~ N(0, 0.9^2) i.i.d. (pre-generated on
the device).  One "step" = one `dm_batch_step` launch = one DPEnv.step (one RK4 mj_step, h = 0.0166 s) of every env
of the rank.  With N > 1 the env index range is sharded over ranks (weak scaling, no per-step collective) and every
256 steps the [256, 4096, 87] f32 rollout block is all-gathered over RCCL, as the learner would consume it — asynchronously,
double-buffered, so the envs keep stepping while the block travels; every gather completes inside the timed region.
Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# the host driver only supports dmabuf IPC: without this, RCCL's cross-process buffer sharing fails (hipIpcGetMemHandle)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ENVS_PER_GPU = 4096
HORIZON = 256
ALGO_BYTES_PER_STEP = 2353      # fp64 state/action in + state/obs/reward/done out, per env-step
ALGO_FLOP_PER_STEP = 1.5e6      # SURVEY.md section 8(d) estimate with ~8 floor contacts + limits
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6    # MI355X fp64 vector peak (public spec)


def cpu_baseline(clip, reward="imitation", budget_s=12.0):
    """Time the CPU oracle (oracle/, float64 C, OpenMP over envs) on a bounded sample of the same workload.  The host may
    expose more logical CPUs than the container can use, so a few thread counts are tried and the best one is reported
    together with the one-core figure."""
    from oracle import oracle as O
    from deepmimic_mujoco_amd import MocapDM, CompiledModel, humanoid_spec
    from deepmimic_mujoco_amd.imitation import ImitationSpec
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    mc = MocapDM(); mc.load_mocap(clip)
    F = mc.data_config.shape[0]
    om = O.Model()
    rng = np.random.RandomState(0)
    imit = reward == "imitation"
    if imit:
        spec = ImitationSpec(CompiledModel(humanoid_spec()))
        table = spec.build_table(mc.data_config, mc.data_vel); params = spec.params(mc.data_config, mc.loop)

    def run(nthreads, seconds):
        n = max(8, nthreads * 8)
        ds = [O.Data(om) for _ in range(n)]
        idx = (np.arange(n) % F).astype(np.int32); cyc = np.zeros(n, dtype=np.int32)
        for e, d in enumerate(ds):
            d.reset(); d.set_state(mc.data_config[e % F], mc.data_vel[e % F])
        steps = 0
        t0 = time.perf_counter()
        while True:
            a = rng.randn(n, 28) * 0.9
            if imit:
                _o, _r, done = O.batch_step_imitation(om, ds, a, 1, table, params, idx, cyc, nthreads)
            else:
                _o, _r, done = O.batch_step(om, ds, a, 1, nthreads)
            steps += 1
            for e in np.nonzero(done)[0]:
                k = rng.randint(F)
                ds[e].reset(); ds[e].set_state(mc.data_config[k], mc.data_vel[k]); idx[e] = k; cyc[e] = 0
            el = time.perf_counter() - t0
            if el > seconds and steps >= 4:
                return n * steps / el, n, steps, el

    counts = sorted({1, min(8, avail), min(32, avail), min(64, avail), avail})
    per = budget_s / len(counts)
    results = {c: run(c, per) for c in counts}
    best = max(results, key=lambda c: results[c][0])
    v, n, steps, el = results[best]
    return {"value": round(v, 1), "unit": "env-steps/s", "cores": best, "kind": "port", "single_core_value": round(results[1][0], 1),
            "by_threads": {str(c): round(results[c][0], 1) for c in counts}, "logical_cpus": avail,
            "sample": "%d envs x %d steps of the same workload (%s, contacts+limits, %s reward, N(0,0.9^2) actions, RSI reset on done), "
                      "oracle/dm_oracle.c fp64 with OpenMP over envs, %d threads (best of %s), %.1f s" % (n, steps, clip, reward, best, counts, el)}


def rollout_bench(args, dev, rank, world, local_rank):
    """Informational: the learner-facing loop of src/trpo.py:27-94 kept on the device — 2x100 tanh policy + value forward,
    Gaussian sampling, env kernel, segment bookkeeping, GAE per 256-step segment.  Not the judged metric line."""
    import torch
    from deepmimic_mujoco_amd import DPVecEnv, MlpPolicy, traj_segment_generator, add_vtarg_and_adv
    n = args.envs
    env = DPVecEnv(n, motion=args.clip, device=local_rank, reward="alive", autoreset="init", seed=0, env_offset=rank * n)
    pol = MlpPolicy(device=dev, seed=0); pol.seed(rank)
    gen = traj_segment_generator(pol, env, HORIZON, stochastic=True)
    segs = max(1, args.steps // HORIZON)
    for _ in range(max(1, args.warmup // HORIZON)):
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
                          "episodes": eps, "config": {"workload": "rollout: %d envs/GPU, untrained 2x100 tanh policy, alive reward, noisy-init autoreset, "
                                                                  "%d-step segments + GAE(0.995, 0.97)" % (n, HORIZON)}}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=512)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--envs", type=int, default=ENVS_PER_GPU, help="envs per GPU")
    ap.add_argument("--workload", default="cfg3", choices=["cfg2", "cfg3", "rollout"],
                    help="cfg3 (default, the judged line) / cfg2: BASELINE.json configs; rollout: policy-in-the-loop segments + GAE (informational)")
    ap.add_argument("--clip", default="walk")
    ap.add_argument("--reward", default="imitation", choices=["imitation", "v3-config", "alive"],
                    help="cfg3 reward: the 5-term DeepMimic imitation reward (default), dp_env_v3's config reward, or the constant 1.0")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--prewarm-horizons", type=int, default=6, help="untimed 256-step horizons before the warm-up steps (cold-box clock ramp, ~1 s)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from deepmimic_mujoco_amd import DPVecEnv, _abi as A

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    n = args.envs
    if args.workload == "rollout":
        return rollout_bench(args, dev, rank, world, local_rank)
    full = args.workload == "cfg3"
    env = DPVecEnv(n, motion=args.clip, device=local_rank, reward=args.reward if full else "alive",
                   autoreset="rsi", seed=0, contacts=full, limits=full,
                   action_mode="raw" if full else "p-control", env_offset=rank * n)
    stream = torch.cuda.Stream(device=dev)
    env.batch.set_stream(stream.cuda_stream)

    with torch.cuda.stream(stream):
        gen = torch.Generator(device=dev); gen.manual_seed(1234 + rank)
        pool = 32
        if full:
            actions = torch.randn((pool, n, A.NU), generator=gen, device=dev, dtype=torch.float64) * 0.9
        else:
            actions = torch.zeros((pool, n, A.NU), device=dev, dtype=torch.float64)   # cfg2: pure P-controller
        # the kernel writes obs / reward / done of step t straight into row t of [T, n, .] staging buffers (no per-step copy
        # kernels); at the end of each 256-step horizon they are packed into the f32 rollout block (obs 56 + act 28 + rew + done
        # + vpred) in one go.  Blocks are double-buffered: while one is all-gathered over RCCL (async, on the collective's own
        # stream) the envs keep stepping
        from deepmimic_mujoco_amd.rollout import DoubleBufferedGather
        obs_T = torch.zeros((HORIZON, n, A.NOBS), dtype=torch.float64, device=dev)    # zeros: every page is touched before the clock starts
        rew_T = torch.zeros((HORIZON, n), dtype=torch.float64, device=dev)
        done_T = torch.zeros((HORIZON, n), dtype=torch.uint8, device=dev)
        tidx = torch.arange(HORIZON, device=dev)
        dbg = DoubleBufferedGather(HORIZON, n, device=dev, world=world)
        env.reset("rsi")

        def one_step(t):
            k = t % HORIZON
            env.batch.step(actions[t % pool], 1, (obs_T[k], rew_T[k], done_T[k]))
            if k == HORIZON - 1:
                blk = dbg.block(t)
                blk[:, :, :56] = obs_T; blk[:, :, 56:84] = actions[(tidx + (t - k)) % pool]
                blk[:, :, 84] = rew_T; blk[:, :, 85] = done_T
                dbg.commit(t)

        drain = dbg.drain

        # untimed: bring a cold box (first process after boot: idle clocks, unmapped VRAM) to its steady state, then the W warm-up steps
        # (a fixed number of horizons, so that every rank issues the same sequence of collectives)
        for _ in range(args.prewarm_horizons):
            for t in range(HORIZON):
                one_step(t)
            drain(); stream.synchronize()
        for t in range(args.warmup):
            one_step(t)
        drain()
        stream.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
        done_count = torch.zeros((), dtype=torch.int64, device=dev)
        t0 = time.perf_counter()
        ev0.record(stream)
        for t in range(args.steps):
            one_step(t)
        drain()                                                            # outstanding gathers finish inside the timed region
        ev1.record(stream)
        stream.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t1 = time.perf_counter()
    elapsed = t1 - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    gpu_ms = ev0.elapsed_time(ev1)
    status = env.batch.get(A.F_STATUS)
    nefc = env.batch.get(A.F_NEFC)

    if rank == 0:
        total_steps = world * n * args.steps
        value = total_steps / elapsed
        kernel_ms = gpu_ms / args.steps       # per-launch duration on the launch stream (incl. k_order and the per-horizon block packing)
        ach_gbs = ALGO_BYTES_PER_STEP * n / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "env-steps/sec", "value": round(value, 1), "unit": "env-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("BASELINE.json configs[2]: 'walk' mocap, %d envs/GPU, full contact + joint-limit PGS solve, "
                                    "%s reward, RSI auto-reset" % (n, "5-term DeepMimic imitation" if args.reward == "imitation" else args.reward)) if full else
                                   ("BASELINE.json configs[1]: 'walk' mocap, %d envs/GPU, P-controller torque, contacts and limits off" % n),
                       "envs_per_gpu": n, "global_envs": world * n, "clip": args.clip, "parallelism": "env-shard x%d" % world,
                       "rollout_allgather_every": HORIZON if world > 1 else None,
                       "mean_nefc": round(float(nefc.mean()), 2), "overflow_envs": int((status & 1).sum())},
            "roofline": {"bound": "hbm", "achieved": round(ach_gbs, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach_gbs / HBM_PEAK_GBS, 6), "traffic": None,
                         "kernel": "k_step_narrow", "kernel_ms": round(kernel_ms, 4),
                         "kernel_ms_covers": "one dm_batch_step on the launch stream: k_step_narrow + k_order (~0.01 ms) + 1/256 of a horizon's block packing", "algorithmic_bytes_per_env_step": ALGO_BYTES_PER_STEP,
                         "note": "latency/ALU-bound path: see fp64 fraction",
                         "fp64_est_tflops": round(ALGO_FLOP_PER_STEP * n / (kernel_ms * 1e-3) / 1e12, 3),
                         "fp64_valu_peak_tflops": FP64_VALU_PEAK_TFLOPS,
                         "fp64_frac": round(ALGO_FLOP_PER_STEP * n / (kernel_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS, 4)},
        }
        tp = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")
        if full and os.path.exists(tp):   # PMC HBM bytes per launch of this kernel/workload, measured by tools/collect_profile.sh (separate rocprofv3 passes)
            try:
                tj = json.load(open(tp))
                out["roofline"]["traffic"] = tj["hbm_bytes_per_launch"]
                out["roofline"]["traffic_source"] = "profiles/r01_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; not re-measured in this run)"
            except Exception:
                pass
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(args.clip, args.reward if full else "alive")
            except Exception as e:  # the baseline must never take the GPU number down with it
                out["cpu_baseline"] = {"value": None, "unit": "env-steps/s", "cores": os.cpu_count(), "kind": "port", "sample": "failed: %r" % (e,)}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
