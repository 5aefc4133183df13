"""Statistical anchors of the (parity-unpinned) physics — TEST INFRASTRUCTURE, CPU, built on the oracle.

The reference ships ONE artefact that was produced inside real MuJoCo 2.0: the TRPO checkpoint
`src/checkpoint_tmp/DeepMimic/trpo-walk-0` (fixture: tests/golden/ckpt/) and the log of the run that wrote it
(tests/golden/trpo_walk0_log.npz).  Its observation filter `pi/obfilter/{runningsum, runningsumsq, count}` holds the first two
moments of all 56 observation dimensions over 11 673 600 samples.  Reading the reference's learner (src/trpo.py:228-296,
338-353: timesteps_per_batch 256, g_step 3, vf_iters 3, minibatch 128, MPI-summed RunningMeanStd, save_per_iter 100), that
count is EXACTLY 1 900 iterations x 3 g-steps x 2 workers x (256 rollout samples + 3 x 256 value-fit samples): the moments are
the mixture of every observation the 2-worker run saw from its untrained start to iteration 1 900 — not the final policy's
stationary distribution.  Two comparisons follow from that:

  * `run_reference_protocol`: the same protocol (2 envs = 2 workers, 256-step segments, the reference's learner settings and
    seeds) run in the oracle's physics with this repository's learner; its final obs-filter moments and its EpLenMean curve
    are compared with the checkpoint's 112 numbers and the reference's log.  Run-to-run spread comes from several seeds.
  * `shipped_policy_moments`: the shipped policy's stationary observation moments (stochastic, src/trpo.py:27-80 protocol)
    — the same statistic on the HIP kernel (tests/test_gpu_rollout.py) must agree with it; against the checkpoint's mixture
    it is only an order-of-magnitude check.
"""
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CKPT = os.path.join(GOLD, "ckpt", "trpo-walk-0")


def checkpoint_moments():
    """(mean[56], std[56], count) of the reference checkpoint's observation filter (src/utils/misc_util.py:53-54)."""
    from deepmimic_mujoco_amd.tf_checkpoint import load_checkpoint
    d = load_checkpoint(CKPT, scope="pi")
    cnt = float(d["obfilter/count"])
    mean = np.asarray(d["obfilter/runningsum"], dtype=np.float64) / cnt
    var = np.asarray(d["obfilter/runningsumsq"], dtype=np.float64) / cnt - mean ** 2
    return mean, np.sqrt(np.maximum(var, 1e-2)), cnt


def moments_of(rms):
    cnt = float(rms.count)
    mean = rms.sum.cpu().numpy() / cnt
    var = rms.sumsq.cpu().numpy() / cnt - mean ** 2
    return mean, np.sqrt(np.maximum(var, 1e-2)), cnt


class _OracleBatch(object):
    def __init__(self, env):
        self.env = env

    def step(self, action, n_substeps=1, out=None):
        return self.env._step(np.asarray(action, dtype=np.float64), n_substeps, out)


class OracleVecEnv(object):
    """`DPVecEnv(autoreset="init")` look-alike on the CPU oracle: N independent `dmo_data`, alive reward, on `done` the
    reference's `env.reset(); env.reset_model_init()` (sim.reset() + noisy default pose from the env's np_random)."""

    def __init__(self, num_envs, seed=0, nthreads=1, **model_opts):
        from oracle import oracle as O
        self.O = O
        self.num_envs = int(num_envs)
        # `spec_*` options scale / replace entries of the model SPECIFICATION before it is compiled (negative controls of the anchors):
        # spec_solref0 / spec_solref1 (replace), spec_gear / spec_damping / spec_friction / spec_armature (scale factors)
        spec_opts = {k: v for k, v in model_opts.items() if k.startswith("spec_")}
        if spec_opts:
            sp = O.humanoid_spec()
            for k, v in spec_opts.items():
                if k in ("spec_solref0", "spec_solref1"):
                    sp.solref[int(k[-1])] = v
                elif k == "spec_gear":
                    for i in range(sp.nu): sp.act_gear[i] *= v
                elif k == "spec_damping":
                    for i in range(sp.njnt): sp.jnt_damping[i] *= v
                elif k == "spec_armature":
                    for i in range(sp.njnt): sp.jnt_armature[i] *= v
                elif k == "spec_friction":
                    for i in range(sp.ngeom): sp.geom_friction[i][0] *= v
                else:
                    raise KeyError(k)
            self.om = O.Model(sp)
        else:
            self.om = O.Model()
        for k, v in model_opts.items():
            if not k.startswith("spec_"):
                self.om.set(k, v)
        self.ds = [O.Data(self.om) for _ in range(self.num_envs)]
        # one np_random per env = per MPI worker of the reference (workerseed = seed + 10000 rank, src/trpo.py:341-343)
        self.rngs = [np.random.RandomState(seed + 10000 * e) for e in range(self.num_envs)]
        self.nthreads = nthreads
        self.batch = _OracleBatch(self)
        self.frame_skip = 1
        self.q0 = self.om.get("qpos0")[:35].copy()

    def _init_pose(self, e):
        d = self.ds[e]; r = self.rngs[e]
        d.reset()
        d.set_state(self.q0 + r.uniform(low=-0.01, high=0.01, size=35), r.uniform(low=-0.01, high=0.01, size=34))

    def _obs(self, e):
        d = self.ds[e]
        return np.concatenate([d.get("qpos")[7:35], d.get("qvel")[6:34]])

    def reset(self, mode="init", out=None):
        for e in range(self.num_envs):
            self._init_pose(e)
        ob = np.stack([self._obs(e) for e in range(self.num_envs)])
        if out is not None:
            out[...] = ob
            return out
        return ob

    def _step(self, action, n_substeps, out):
        obs, rew, done = self.O.batch_step(self.om, self.ds, action.reshape(self.num_envs, 28), int(n_substeps), self.nthreads)
        for e in np.nonzero(done)[0]:
            self._init_pose(e)
            obs[e] = self._obs(e)
        if out is not None:
            out[0][...] = obs; out[1][...] = rew; out[2][...] = done
            return out
        return obs, rew, done

    def close(self):
        pass


def run_reference_protocol(seed=0, iterations=1900, workers=2, horizon=256, g_step=3, log_every=0, backend="oracle", **model_opts):
    model_opts = {("iterations" if k == "pgs_iterations" else k): v for k, v in model_opts.items()}      # (the oracle's PGS sweep limit shares a name with the protocol's)
    """The reference's training run (`mpirun -np 2 python3 trpo.py`: src/trpo.py:338-353) in the oracle's physics
    (backend="oracle", CPU) or on the HIP kernel (backend="gpu": a 2-env `DPVecEnv` with the kernel's noisy-init auto-reset).
    Returns {"EpLenMean": per-iteration curve (rolling 40 episodes, logged every g_step updates like src/trpo.py:303-306),
    "TimestepsSoFar", "mean", "std", "count"}."""
    import torch
    from collections import deque
    from deepmimic_mujoco_amd.policy import MlpPolicy
    from deepmimic_mujoco_amd.rollout import traj_segment_generator
    from deepmimic_mujoco_amd.trpo import TrpoLearner
    torch.manual_seed(seed); torch.set_num_threads(1)
    if backend == "gpu":
        from deepmimic_mujoco_amd import DPVecEnv
        assert not model_opts, "the kernel has no model switches"
        env = DPVecEnv(workers, motion="walk", device=0, reward="alive", autoreset="init", seed=seed)
        pi = MlpPolicy(device="cuda:0", seed=seed); pi.seed(seed)
    else:
        env = OracleVecEnv(workers, seed=seed, **model_opts)
        pi = MlpPolicy(device="cpu", seed=seed); pi.seed(seed)
    # value-fit minibatch: each worker walks its own 256 samples in minibatches of 128 and the gradients are all-mean'd,
    # i.e. `workers` x 128 samples per Adam step (src/trpo.py:288-295)
    learner = TrpoLearner(pi, vf_batch_size=128 * workers, seed=seed)
    gen = traj_segment_generator(pi, env, horizon, stochastic=True, first_reset="init")
    lenbuf = deque(maxlen=40)
    curve, steps, tot = [], [], 0
    for it in range(iterations):
        for g in range(g_step):
            seg = next(gen)
            learner.update(seg)
        lens = seg["ep_lens"]                          # the last g-step's episodes are the ones logged (src/trpo.py:298-309)
        lenbuf.extend(lens); tot += int(sum(lens))
        curve.append(float(np.mean(lenbuf)) if lenbuf else float("nan")); steps.append(tot)
        if log_every and (it + 1) % log_every == 0:
            print("seed %d iter %d EpLenMean %.1f steps %d" % (seed, it + 1, curve[-1], tot), flush=True)
    mean, std, cnt = moments_of(pi.ob_rms)
    return {"seed": seed, "iterations": iterations, "EpLenMean": curve, "TimestepsSoFar": steps, "mean": mean.tolist(), "std": std.tolist(),
            "count": cnt, "entropy": float(pi.entropy()), "model_opts": model_opts, "backend": backend}


def shipped_policy_moments(n=64, steps=400, seed=0, nthreads=None, **model_opts):
    """Stationary observation moments of the SHIPPED policy (stochastic) under the trainer's episode protocol, on the oracle.
    -> (mean[56], std[56], samples, mean first-episode length)."""
    import torch
    from deepmimic_mujoco_amd.policy import MlpPolicy
    pol = MlpPolicy.from_tf_checkpoint(CKPT); pol.seed(seed)
    env = OracleVecEnv(n, seed=seed, nthreads=nthreads or min(n, os.cpu_count() or 1), **model_opts)
    ob = env.reset("init")
    s = np.zeros(56); s2 = np.zeros(56); cnt = 0
    first = np.full(n, -1)
    for t in range(steps):
        s += ob.sum(0); s2 += (ob * ob).sum(0); cnt += n
        ac, _ = pol.act(True, torch.from_numpy(ob))
        ob, _r, done = env._step(ac.numpy(), 1, None)
        newly = (first < 0) & (done != 0)
        first[newly] = t + 1
    mean = s / cnt
    return mean, np.sqrt(np.maximum(s2 / cnt - mean ** 2, 1e-2)), cnt, float(np.where(first < 0, steps, first).mean())


def compare_moments(mean, std, ref_mean=None, ref_std=None):
    """Per-dimension deviation of (mean, std) from the checkpoint's moments, in units of the checkpoint's std:
    dmean = (mean - ref_mean) / ref_std,  rstd = std / ref_std."""
    if ref_mean is None:
        ref_mean, ref_std, _ = checkpoint_moments()
    mean = np.asarray(mean); std = np.asarray(std)
    return (mean - ref_mean) / ref_std, std / ref_std
