"""Multi-GPU sharding of the environment index range and the per-horizon rollout exchange.

Environments are independent, so the only collective is the one the learner needs: every `horizon` steps each rank
contributes its [T, N_local, 87] float32 block (obs 56 | action 28 | reward | done | vpred) to an all-gather
(RCCL over xGMI when the backend is "nccl"; "gloo" in the CPU tests).  Env e of the global range lives on rank
e // N_local, and per-env RNG streams are keyed by the GLOBAL env id, so results do not depend on the sharding.
"""
import numpy as np

ROW = 87  # 56 obs + 28 act + reward + done + vpred


def shard_range(n_global, rank, world):
    """Contiguous block [lo, hi) of the global env range owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(int(n_global), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class RolloutBlock(object):
    """Accumulates T steps of (obs, action, reward, done[, vpred]) for the local shard; all-gathers on demand."""

    def __init__(self, horizon, n_local, device="cpu"):
        import torch
        self.T, self.n = int(horizon), int(n_local)
        self.buf = torch.zeros((self.T, self.n, ROW), dtype=torch.float32, device=device)
        self.t = 0

    def append(self, obs, action, reward, done, vpred=None):
        row = self.buf[self.t]
        row[:, :56] = obs; row[:, 56:84] = action; row[:, 84] = reward; row[:, 85] = done
        if vpred is not None:
            row[:, 86] = vpred
        self.t += 1
        return self.t == self.T

    def gather(self, out=None):
        """All ranks receive every rank's block: returns [world, T, n_local, 87].  Resets the fill counter."""
        import torch
        import torch.distributed as dist
        self.t = 0
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return self.buf.unsqueeze(0)
        world = dist.get_world_size()
        if out is None:
            out = torch.empty((world,) + tuple(self.buf.shape), dtype=self.buf.dtype, device=self.buf.device)
        dist.all_gather_into_tensor(out.view(world * self.T, self.n, ROW), self.buf)
        return out


# ---- device-resident rollouts -----------------------------------------------------------

def add_vtarg_and_adv(seg, gamma, lam):
    """GAE(lambda) of `src/trpo.py:83-94` for N environments at once.

    seg["rew"], seg["vpred"], seg["new"] are [T, N]; seg["nextvpred"] is [N].  Per env column this is the reference's
    reversed loop verbatim:  nonterminal = 1 - new[t+1];  delta = rew[t] + gamma * vpred[t+1] * nonterminal - vpred[t];
    adv[t] = delta + gamma * lam * nonterminal * adv[t+1];  tdlamret = adv + vpred  — float32 like the reference's
    `np.empty(T, 'float32')`.  T sequential steps of N-wide vector ops on the tensors' device (no host round trip)."""
    import torch
    rew, vpred, new = seg["rew"], seg["vpred"], seg["new"]
    T = rew.shape[0]
    new1 = torch.cat([new.to(torch.float32), torch.zeros_like(new[:1], dtype=torch.float32)], 0)     # np.append(new, 0)
    vp1 = torch.cat([vpred.to(torch.float32), seg["nextvpred"].to(torch.float32)[None]], 0)          # np.append(vpred, nextvpred)
    nonterminal = 1.0 - new1[1:]
    delta = rew.to(torch.float32) + gamma * vp1[1:] * nonterminal - vp1[:-1]
    decay = (gamma * lam) * nonterminal
    adv = torch.empty_like(delta)
    last = torch.zeros_like(delta[0])
    for t in range(T - 1, -1, -1):
        last = delta[t] + decay[t] * last
        adv[t] = last
    seg["adv"] = adv
    seg["tdlamret"] = adv + vpred.to(torch.float32)
    return seg


def traj_segment_generator(pi, env, horizon, stochastic=True, device=None, first_reset="rsi"):
    """Batched `traj_segment_generator` (src/trpo.py:27-80): N envs advance in lock step on the device.

    pi: policy.MlpPolicy; env: DPVecEnv created with autoreset="init" — the kernel then applies, on `done`, exactly what
    the reference does on the host (`env.reset(); ob = env.env.reset_model_init()`, :77-79) and returns the fresh
    episode's observation.  Yields, every `horizon` steps, the reference's segment dict with a leading [T, N] shape:
    ob [T,N,56] f32, ac / prevac [T,N,28] f32, rew / vpred [T,N] f32, new [T,N] int32 (new[t] = ob[t] starts an episode),
    nextvpred [N], ep_rets / ep_lens (lists of finished episodes, host ints — the only host transfer, once per segment).
    Observations, actions, rewards and dones never leave the device; policy and env share the current torch stream."""
    import torch
    n = env.num_envs
    if device is None:
        device = pi.device
    f32, f64 = torch.float32, torch.float64
    obs = torch.zeros((horizon, n, 56), dtype=f32, device=device)
    acs = torch.zeros((horizon, n, 28), dtype=f32, device=device)
    prevacs = torch.zeros_like(acs)
    rews = torch.zeros((horizon, n), dtype=f32, device=device)
    vpreds = torch.zeros((horizon, n), dtype=f32, device=device)
    news = torch.zeros((horizon, n), dtype=torch.int32, device=device)
    ep_ret_log = torch.zeros((horizon, n), dtype=f64, device=device)      # return / length of the episode that ended at (t, e)
    ep_len_log = torch.zeros((horizon, n), dtype=torch.int32, device=device)

    ob = torch.empty((n, 56), dtype=f64, device=device)
    rew = torch.empty(n, dtype=f64, device=device)
    done = torch.empty(n, dtype=torch.uint8, device=device)
    ac = torch.zeros((n, 28), dtype=f64, device=device)
    prevac = torch.zeros((n, 28), dtype=f64, device=device)
    new = torch.ones(n, dtype=torch.int32, device=device)
    cur_ret = torch.zeros(n, dtype=f64, device=device)
    cur_len = torch.zeros(n, dtype=torch.int32, device=device)
    as_buf = (lambda x: x) if torch.device(device).type == "cuda" else (lambda x: x.numpy())   # host tensors: shared-memory views
    step_out = (as_buf(ob), as_buf(rew), as_buf(done))
    env.reset(first_reset, out=as_buf(ob))                                # trpo.py:32 `ob = env.reset()` (RSI); later episodes: noisy init
    t = 0
    while True:
        prevac.copy_(ac)
        _, vpred = pi.act(stochastic, ob, out=ac)
        if t > 0 and t % horizon == 0:
            ended = ep_len_log > 0
            # time-major order = the order in which a single-env loop would have appended them
            ep_rets = ep_ret_log[ended].tolist()
            ep_lens = ep_len_log[ended].tolist()
            yield {"ob": obs, "rew": rews, "vpred": vpreds, "new": news, "ac": acs, "prevac": prevacs,
                   "nextvpred": vpred * (1 - new).to(f32), "ep_rets": ep_rets, "ep_lens": ep_lens}
            ep_ret_log.zero_(); ep_len_log.zero_()
        i = t % horizon
        obs[i] = ob; vpreds[i] = vpred; news[i] = new; acs[i] = ac; prevacs[i] = prevac
        env.batch.step(as_buf(ac), 1, step_out)                          # :66 `env.step(ac)` for every env, one launch
        rews[i] = rew
        cur_ret += rew; cur_len += 1
        new = done.to(torch.int32)
        fin = done.bool()
        ep_ret_log[i] = torch.where(fin, cur_ret, torch.zeros_like(cur_ret))
        ep_len_log[i] = torch.where(fin, cur_len, torch.zeros_like(cur_len))
        cur_ret.masked_fill_(fin, 0.0); cur_len.masked_fill_(fin, 0)
        t += 1


def flatten_segment(seg):
    """[T, N, ...] -> [T*N, ...] views env-major (each env's T steps contiguous), the layout the reference's learner
    consumes when several workers' segments are concatenated."""
    out = {}
    for k in ("ob", "ac", "prevac", "rew", "vpred", "new", "adv", "tdlamret"):
        if k in seg:
            v = seg[k]
            out[k] = v.transpose(0, 1).reshape((-1,) + tuple(v.shape[2:]))
    return out
