#!/usr/bin/env python3
"""Train the humanoid with the TRPO learner on device-resident rollouts (the reference's `python3 trpo.py`, src/trpo.py:438-491).

    python tools/train_trpo.py --envs 1024 --horizon 64 --seconds 120 [--out gpurun_out/trpo_curve.json]
    torchrun --nproc-per-node 8 --master-addr 127.0.0.1 tools/train_trpo.py ...      (one rank per GPU, all-mean'd updates)
"""
import argparse
import json
import os
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only (RCCL across processes)
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from deepmimic_mujoco_amd import DPVecEnv, MlpPolicy  # noqa: E402
from deepmimic_mujoco_amd.trpo import learn  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=1024)
    ap.add_argument("--horizon", type=int, default=64)
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--iters", type=int, default=0)
    ap.add_argument("--vf-batch", type=int, default=4096)
    ap.add_argument("--vf-stepsize", type=float, default=1e-3)
    ap.add_argument("--max-kl", type=float, default=0.01)
    ap.add_argument("--motion", default="walk")
    ap.add_argument("--reward", default="alive", help="alive | v3-config | v2-pose | imitation")
    ap.add_argument("--autoreset", default="init", help="init (the reference's trpo.py protocol) | rsi (DeepMimic reference-state initialisation)")
    ap.add_argument("--frame-skip", default=None, help="sim steps per env step, or 'mocap' (default: 1; 'mocap' with --reward imitation)")
    ap.add_argument("--pipeline", type=int, default=2, help="sub-batches whose step launches overlap across consecutive steps (DM_OPT_PIPELINE; with "
                                                            "--unfused: that many env batches on their own streams, policy -> env chains overlap)")
    ap.add_argument("--unfused", action="store_true", help="separate policy launch per step instead of the policy step inside the env step kernel")
    ap.add_argument("--task", default="train", choices=["train", "evaluate"], help="evaluate: the reference's `trpo.py --task evaluate --load_model_path ...`")
    ap.add_argument("--load-model-path", default=None, help="evaluate: a tf.train.Saver checkpoint prefix (the reference's or one written by --save) or an .npz")
    ap.add_argument("--number-trajs", type=int, default=10, help="evaluate: trajectories (one env each; src/trpo.py:483)")
    ap.add_argument("--stochastic-policy", action="store_true")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default=None)
    ap.add_argument("--log-dir", default=None, help="write progress.csv and monitor.csv in the reference's formats")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"], help="nccl = RCCL, one GPU per rank; gloo = ranks may share a GPU (LOCAL_RANK modulo the visible devices)")
    ap.add_argument("--no-pg-native", action="store_true", help="policy half of the update through torch autograd instead of csrc/pg_kernel.h")
    ap.add_argument("--dump-params", default=None, help="every rank writes its final parameters to <prefix>.rank<r>.npz (replica-consistency checks)")
    ap.add_argument("--save", default=None, help="write the trained policy: `x.npz` (reference variable names) or a checkpoint prefix -> "
                                                 "tf.train.Saver bundle (x.index + x.data-00000-of-00001) the reference's `--task evaluate --load_model_path x` restores")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0")); lr = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    if args.dist_backend == "nccl" and world > ndev:
        raise SystemExit("RCCL needs one GPU per rank: %d ranks, %d devices visible (use --dist-backend gloo to share a GPU)" % (world, ndev))
    lr = lr % max(1, ndev)
    torch.cuda.set_device(lr)
    dev = torch.device("cuda", lr)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    fs = args.frame_skip if args.frame_skip in (None, "mocap") else int(args.frame_skip)
    if args.task == "evaluate":                     # src/trpo.py:480-487
        from deepmimic_mujoco_amd.trpo import runner
        assert args.load_model_path, "--task evaluate needs --load-model-path"
        pi = MlpPolicy.from_npz(args.load_model_path, device=dev) if args.load_model_path.endswith(".npz") else MlpPolicy.from_tf_checkpoint(args.load_model_path, device=dev)
        pi.seed(args.seed)
        env = DPVecEnv(args.number_trajs, motion=args.motion, device=lr, reward=args.reward, autoreset="init", seed=args.seed, frame_skip=fs)
        runner(env, pi, timesteps_per_batch=1024, stochastic_policy=args.stochastic_policy)
        return
    P = max(1, args.pipeline)
    if args.unfused:
        cuts = [args.envs * h // P for h in range(P + 1)]
        envs = [DPVecEnv(cuts[h + 1] - cuts[h], motion=args.motion, device=lr, reward=args.reward, autoreset=args.autoreset, seed=args.seed + 10000 * rank,
                         env_offset=rank * args.envs + cuts[h], frame_skip=fs) for h in range(P)]
        env = envs if P > 1 else envs[0]
    else:
        from deepmimic_mujoco_amd import _abi as A
        env = DPVecEnv(args.envs, motion=args.motion, device=lr, reward=args.reward, autoreset=args.autoreset, seed=args.seed + 10000 * rank,
                       env_offset=rank * args.envs, frame_skip=fs)
        env.batch.set_option(A.OPT_PIPELINE, min(P, A.MAX_PIPELINE))
    pi = MlpPolicy(device=dev, seed=args.seed); pi.seed(args.seed + 10000 * rank)
    hist = learn(env, pi, timesteps_per_batch=args.horizon, max_seconds=args.seconds if not args.iters else 0, max_iters=args.iters,
                 vf_batch_size=args.vf_batch, vf_stepsize=args.vf_stepsize, max_kl=args.max_kl, seed=args.seed, log_dir=args.log_dir,
                 fused=False if args.unfused else None, pg_native=False if args.no_pg_native else None)
    if args.dump_params:
        os.makedirs(os.path.dirname(os.path.abspath(args.dump_params)), exist_ok=True)
        pi.save_npz("%s.rank%d.npz" % (args.dump_params, rank))
    if rank == 0:
        if args.out:
            os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
            json.dump({"args": vars(args), "world": world, "history": hist}, open(args.out, "w"))
        if args.save:
            if args.save.endswith(".npz"):
                pi.save_npz(args.save)
            else:
                pi.save_tf_checkpoint(args.save)
        best = max(h["EpLenMeanIter"] for h in hist)
        print("done: %d iterations, %d env steps in %.1f s (%.0f steps/s incl. learner), EpLenMean(last iter) %.1f, best %.1f"
              % (len(hist), hist[-1]["TimestepsSoFar"], hist[-1]["TimeElapsed"], hist[-1]["TimestepsSoFar"] / hist[-1]["TimeElapsed"],
                 hist[-1]["EpLenMeanIter"], best))


if __name__ == "__main__":
    main()
