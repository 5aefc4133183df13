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
    if (rew.is_cuda and rew.dim() == 2 and rew.dtype == torch.float32 and vpred.dtype == torch.float32 and new.dtype == torch.int32
            and rew.is_contiguous() and vpred.is_contiguous() and new.is_contiguous()):
        # one launch of the k_gae kernel (csrc/policy_kernel.h) instead of T small ones
        import ctypes as C
        from . import _abi as A
        L = A.load()
        nxt = seg["nextvpred"].to(torch.float32).contiguous()
        adv = torch.empty_like(rew); ret = torch.empty_like(rew)
        p = lambda x: C.c_void_p(x.data_ptr())
        A.check(L.dm_gae(p(rew), p(vpred), p(new), p(nxt), p(adv), p(ret), T, rew.shape[1], float(gamma), float(lam),
                         C.c_void_p(torch.cuda.current_stream(rew.device).cuda_stream)), L)
        seg["adv"], seg["tdlamret"] = adv, ret
        return seg
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


def can_fuse(pi, env, device=None):
    """True when `SegmentCollector(fused=True)` is possible: the policy step can run inside the env step kernel (dm_batch_step_act)."""
    import torch
    device = torch.device(pi.device if device is None else device)
    return (device.type == "cuda" and getattr(pi, "native", False) and hasattr(getattr(env, "batch", None), "step_act")
            and getattr(env.batch, "can_step_act", True) and getattr(pi, "ob_dim", 0) == 56 and getattr(pi, "ac_dim", 0) == 28 and getattr(pi, "hid_size", 0) == 100)


class _PendingEpisodes:
    """Episode statistics of a segment whose records are still on their way to the host (SegmentCollector._episodes_native)."""

    def __init__(self, collector, event):
        self.c, self.event, self.out = collector, event, None

    def result(self):
        if self.out is None:
            import numpy as np
            cnt, rec, h_cnt, h_rec = self.c._ep_buf
            self.event.synchronize()
            k = min(int(h_cnt[0]), rec.shape[0])
            r = h_rec[:min(k, h_rec.shape[0])].numpy()
            if k > h_rec.shape[0]:
                r = np.concatenate([r, rec[h_rec.shape[0]:k].cpu().numpy()], 0)
            order = np.argsort(r[:, 0], kind="stable")                     # word 0 = t << 32 | env: time-major, then env
            self.out = (r[order, 1].copy().view(np.float64).tolist(), r[order, 2].tolist())
            if self.c._ep_pending is self:
                self.c._ep_pending = None
        return self.out


class Segment(dict):
    """The generator's segment dict.  "ep_rets" / "ep_lens" materialise on first access when the collector left them pending
    (`finish_episode_stats()` does it explicitly — the learner calls it where the host would otherwise wait for the device)."""

    pending_episodes = None
    info = None                          # how the segment was produced (launch form, overflow rate of the packed path): diagnostics, not data

    def finish_episode_stats(self):
        if self.pending_episodes is not None:
            rets, lens = self.pending_episodes.result()
            self.pending_episodes = None
            dict.__setitem__(self, "ep_rets", rets); dict.__setitem__(self, "ep_lens", lens)

    def __missing__(self, key):
        if key in ("ep_rets", "ep_lens") and self.pending_episodes is not None:
            self.finish_episode_stats()
            return dict.__getitem__(self, key)
        raise KeyError(key)

    def __contains__(self, key):
        return dict.__contains__(self, key) or (key in ("ep_rets", "ep_lens") and self.pending_episodes is not None)

    # whoever walks the dict sees all of it
    def __iter__(self):
        self.finish_episode_stats(); return dict.__iter__(self)

    def __len__(self):
        self.finish_episode_stats(); return dict.__len__(self)

    def keys(self):
        self.finish_episode_stats(); return dict.keys(self)

    def items(self):
        self.finish_episode_stats(); return dict.items(self)

    def values(self):
        self.finish_episode_stats(); return dict.values(self)

    def get(self, key, default=None):
        return self[key] if key in self else default


class SegmentCollector(object):
    """One env batch's side of `traj_segment_generator`, split into `launch()` (enqueue T policy + env steps, no host wait) and
    `collect()` (episode bookkeeping, the one host transfer per segment) so that several env batches can be in flight at once.
    With `stream` (a torch CUDA stream) everything this collector enqueues runs on that stream."""

    def __init__(self, pi, env, horizon, stochastic=True, device=None, first_reset="rsi", stream=None, fused=False):
        import torch
        self.pi, self.env, self.T, self.stochastic, self.stream = pi, env, int(horizon), stochastic, stream
        n, T = env.num_envs, self.T
        self.n = n
        device = torch.device(pi.device if device is None else device)
        self.device = device
        f32, f64 = torch.float32, torch.float64
        self.ob64 = torch.zeros((T + 1, n, 56), dtype=f64, device=device)         # row t: observation the policy sees at step t
        self.ac64 = torch.zeros((T + 1, n, 28), dtype=f64, device=device)         # row T: the action already drawn for ob64[T] (fused path)
        self.rew64 = torch.zeros((T, n), dtype=f64, device=device)
        self.done8 = torch.zeros((T, n), dtype=torch.uint8, device=device)
        self.vpreds = torch.zeros((T + 1, n), dtype=f32, device=device)
        self.first = torch.ones(n, dtype=torch.int32, device=device)               # `new` of row 0: carried over from the last segment
        self.last_ac = torch.zeros((n, 28), dtype=f32, device=device)              # prevac of row 0 (trpo.py:29 samples a random one)
        self.cur_ret = torch.zeros(n, dtype=f64, device=device)                    # running return / length of the open episodes
        self.cur_len = torch.zeros(n, dtype=torch.int64, device=device)
        self.as_buf = (lambda x: x) if device.type == "cuda" else (lambda x: x.numpy())   # host tensors: shared-memory views
        self.step_idx = torch.arange(1, T + 1, device=device, dtype=torch.int64)[:, None]
        # fused path: the env step kernel also runs the policy on the observation it produced (dm_batch_step_act), one launch per step.
        # As in the reference's generator (src/trpo.py:47-56) the action for the first observation of a segment was then drawn BEFORE the
        # update in between (by the policy that finished the last segment) and only its value is re-evaluated afterwards.
        if fused and not can_fuse(pi, env, device):
            raise ValueError("fused rollout steps need the native policy kernel (56-100-100-28) and a device-resident env batch")
        self.fused = bool(fused)
        self.have_ac0 = False
        self._redo_seen = None
        import os
        if os.environ.get("DM_HORIZON_REDO_RATE_MAX"):                   # (experiments: tools/prof_train.sh)
            self.HORIZON_REDO_RATE_MAX = float(os.environ["DM_HORIZON_REDO_RATE_MAX"])
        self._packed_now = None                    # what _choose_kernel picked for the last horizon (None: none yet)
        self.kernel_switches = 0
        with self._on_stream():
            env.reset(first_reset, out=self.as_buf(self.ob64[0]))                  # trpo.py:32 `ob = env.reset()` (RSI); later episodes: noisy init

    def _on_stream(self):
        import contextlib
        import torch
        return torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def launch(self):
        import torch
        pi, env, T, as_buf = self.pi, self.env, self.T, self.as_buf
        ob64, ac64, rew64, done8, vpreds = self.ob64, self.ac64, self.rew64, self.done8, self.vpreds
        fs = getattr(env, "frame_skip", 1)
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream(self.device))         # parameters / filter updated by the learner are visible
        if self.fused and (getattr(pi, "_dirty", False) or getattr(pi, "_packed", None) is None):
            pi.pack()                                                               # (on the caller's stream, before the side stream forks)
            if self.stream is not None:
                self.stream.wait_stream(torch.cuda.current_stream(self.device))
        with self._on_stream(), torch.no_grad():                                    # (the learner may hold the parameters with requires_grad)
            if self.fused:
                if not self.have_ac0:
                    pi.act(self.stochastic, ob64[0], out=ac64[0], vpred_out=vpreds[0])                   # very first step: :49
                else:
                    vpreds[0] = pi.forward_value(ob64[0])                                                 # :56, the updated policy's value
                if hasattr(env.batch, "rollout"):
                    # the whole horizon in one call (dm_batch_rollout): on the packed path ONE launch in which every wavefront runs its four
                    # environments through all T steps at its own pace; on the one-env path T step launches issued without returning here
                    restore = self._choose_kernel()
                    try:
                        env.batch.rollout(ac64, (ob64[1:], rew64, done8), fs, pi._packed, vpreds[1:], self.stochastic, pi._seed, pi._counter + 1)
                        pi._counter += T
                        env.batch.join()
                    finally:                                        # (also when the rollout raises: the batch must not be left on the collector's kernel choice)
                        if restore is not None:
                            restore()                               # per-step callers of the same env keep the kernel THEY were given
                    return
                for t in range(T):                                                                       # :49 + :66, one launch
                    pi._counter += 1
                    env.batch.step_act(ac64[t], fs, (ob64[t + 1], rew64[t], done8[t]), pi._packed, ac64[t + 1], vpreds[t + 1],
                                       self.stochastic, pi._seed, pi._counter)
                env.batch.join()
                return
            for t in range(T):
                pi.act(self.stochastic, ob64[t], out=ac64[t], vpred_out=vpreds[t])                       # :49
                env.batch.step(as_buf(ac64[t]), fs, (as_buf(ob64[t + 1]), as_buf(rew64[t]), as_buf(done8[t])))   # :66, one launch
            vpreds[T] = pi.forward(ob64[T])[1]                                      # value of the observation after the segment (:49-52);
        # the action for it is sampled at the top of the next segment, i.e. from the policy as updated in between

    # horizon launch -> one-env steps: an in-wave re-step holds a wave (four environments) for about one lone one-env step.  Measured on a
    # population learning to stand (tools/train_trpo.py, 4 096 envs x 128 steps, MI355X; profiles/r04_ab_kernel_variants.md): 38 / 42 / 46 /
    # 52 / 55 / 58-60 ms per horizon at 0.2 / 0.8 / 1.3 / 2.6 / 3.7 / 5 % of env-steps re-stepped (the rate levels off near 5 %) against
    # 67 ms for the same population through one-env launches — the packed horizon stays ahead to about 7 %.  (Round 3's 2 % came from a cost
    # estimate; a run that crossed it at 2.02 % lost a quarter of its rollout throughput for the rest of the training.)
    HORIZON_REDO_RATE_MAX = 7e-2
    HORIZON_HEAVY_ROWS = 38                # the way back to the packed horizon launch: nobody above this (it holds _abi.PACKED_MAXROWS = 40 rows per env)

    def _choose_kernel(self):
        """Four environments per wavefront or one, for the next horizon (envs that leave the choice open: DPVecEnv.horizon_packed_ok).
        The horizon launch re-steps an environment beyond the packed path's capacities inside its wave, which holds up its three
        partners: fine at the 1e-4 rates of a falling / walking population, not for one that stands on both feet (32+ rows) most of the
        time.  Decided from the last horizon's own statistics (redo rate on the packed path, largest row count on the one-env path):
        a deterministic function of the trajectory."""
        from . import _abi as A
        env, b = self.env, self.env.batch
        if not getattr(env, "horizon_packed_ok", False):
            return None
        # the choice holds for this collector's rollout calls only: what the batch was set to (and its per-step chooser, if on) comes back
        # after each call, so `DPVecEnv.step` callers of the same env are not moved to a kernel that is the slower one per step
        was_mode = int(b.__dict__.get("options", {}).get(A.OPT_PACKED, 0))      # (0 one env per wave, 1 packed, 2 packed with the three-set code per step)
        was_on = bool(was_mode)
        was_auto = bool(b.__dict__.get("_auto"))
        on = getattr(self, "_packed_now", None)
        if on is None:
            on = was_on
        if self._redo_seen is None:                                         # first horizon: start packed
            want = True
        elif on:
            redo = b.redo_total()
            self._last_redo_rate = (redo - self._redo_seen) / float(self.T * self.n)
            want = self._last_redo_rate <= self.HORIZON_REDO_RATE_MAX
        else:
            want = int(b.get(A.F_NEFC).max()) <= self.HORIZON_HEAVY_ROWS
        if want != on:
            self.kernel_switches += 1
        self._packed_now = want
        if was_auto:
            b._auto = False                                                 # (suspended, not reset: its counters go on afterwards)
        if want != was_on:
            b.set_option(A.OPT_PACKED, 1 if want else 0)
        self._redo_seen = b.redo_total()

        def restore():
            if want != was_on:
                b.set_option(A.OPT_PACKED, was_mode)
            if was_auto:
                b._auto = True
                b.rebaseline_auto()                                         # (the horizon's in-wave re-steps are not the per-step chooser's evidence)
        return restore

    def collect(self):
        import torch
        T, n, device = self.T, self.n, self.device
        f32 = torch.float32
        if self.stream is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.stream)
        import os
        import time
        prof = os.environ.get("DM_TRPO_PROFILE") and device.type == "cuda"
        if prof:
            torch.cuda.synchronize(device); t_c = time.perf_counter()
        ob64, ac64, rew64, done8, vpreds = self.ob64, self.ac64, self.rew64, self.done8, self.vpreds
        done = done8.to(torch.bool)
        new = torch.cat([self.first[None], done8[:-1].to(torch.int32)], 0)
        acs = ac64[:T].to(f32)
        prevacs = torch.cat([self.last_ac[None], acs[:-1]], 0)
        # episode statistics: return / length of every episode that ended inside the segment, in time-major order
        # (= the order in which a single-env loop would have appended them, :72-76)
        native_eps = device.type == "cuda" and rew64.dtype == torch.float64 and rew64.is_contiguous() and done8.is_contiguous()
        seg = Segment()
        seg.info = {"packed": self._packed_now, "kernel_switches": self.kernel_switches, "redo_rate": getattr(self, "_last_redo_rate", None)}
        if native_eps:
            seg.pending_episodes = self._episodes_native(rew64, done8)
        else:
            seg["ep_rets"], seg["ep_lens"] = self._episodes_torch(rew64, done, T, n, device)
        seg.update({"ob": ob64[:T].to(f32), "rew": rew64.to(f32), "vpred": vpreds[:T].clone(), "new": new, "ac": acs, "prevac": prevacs,
                    "nextvpred": vpreds[T] * (1 - done8[-1].to(f32))})
        self.first = done8[-1].to(torch.int32)
        self.last_ac = acs[-1].clone()
        ob64[0].copy_(ob64[T])
        if self.fused:
            ac64[0].copy_(ac64[T]); self.have_ac0 = True
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream(self.device))   # the copies above are ordered before the next launch
        if prof:
            torch.cuda.synchronize(device); self.collect_ms = (time.perf_counter() - t_c) * 1e3
            seg["collect_ms"] = self.collect_ms
        return seg

    EP_HEAD = 16384                      # episode records fetched with the count in one asynchronous copy (more than that: a second copy)

    def _episodes_native(self, rew64, done8):
        """One launch (dm_episode_scan: thread = env walks its column of the segment) and a host sort of the few episodes that ended, instead
        of ~100 launch-bound tensor ops.  Returns are float64 sums in step order, like the reference's `cur_ep_ret += rew`.  Nothing waits
        here: the records travel to pinned memory behind the kernel, and `_PendingEpisodes.result()` sorts them when somebody asks — the
        learner does while the device is busy with the update's first kernels."""
        import ctypes as C
        import torch
        from . import _abi as A
        T, n, dev = self.T, self.n, self.device
        if getattr(self, "_ep_buf", None) is None:
            cap = T * n
            head = min(cap, self.EP_HEAD)
            self._ep_buf = (torch.zeros(1, dtype=torch.int32, device=dev), torch.empty((cap, 3), dtype=torch.int64, device=dev),
                            torch.zeros(1, dtype=torch.int32).pin_memory(), torch.empty((head, 3), dtype=torch.int64).pin_memory())
            self._ep_pending = None
        if self._ep_pending is not None:
            self._ep_pending.result()                                      # (its pinned buffers are about to be overwritten)
        cnt, rec, h_cnt, h_rec = self._ep_buf
        p = lambda x: C.c_void_p(x.data_ptr())
        L = A.load()
        A.check(L.dm_episode_scan(p(rew64), p(done8), T, n, p(self.cur_ret), p(self.cur_len), p(cnt), rec.shape[0], p(rec),
                                  C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), L)
        h_cnt.copy_(cnt, non_blocking=True)
        h_rec.copy_(rec[:h_rec.shape[0]], non_blocking=True)
        ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream(dev))
        self._ep_pending = _PendingEpisodes(self, ev)
        return self._ep_pending

    def _episodes_torch(self, rew64, done, T, n, device):
        import torch
        csum = torch.cumsum(rew64, 0)
        ends = done.nonzero()                                              # [K, 2] (t, env), sorted by t then env
        ep_rets, ep_lens = [], []
        cur_ret, cur_len = self.cur_ret, self.cur_len
        if ends.numel():
            te, ee = ends[:, 0], ends[:, 1]
            # previous end of the same env inside the segment (or -1): sort by (env, t) and shift
            order = torch.argsort(ee * T + te)
            te_s, ee_s = te[order], ee[order]
            prev_t = torch.full_like(te_s, -1)
            same = ee_s[1:] == ee_s[:-1]
            prev_t[1:] = torch.where(same, te_s[:-1], torch.full_like(te_s[:-1], -1))
            seg_ret = csum[te_s, ee_s] - torch.where(prev_t >= 0, csum[prev_t.clamp(min=0), ee_s], torch.zeros_like(csum[te_s, ee_s]))
            seg_len = te_s - prev_t
            opening = prev_t < 0                                           # first end of that env: add what it carried in
            seg_ret = seg_ret + torch.where(opening, cur_ret[ee_s], torch.zeros_like(seg_ret))
            seg_len = seg_len + torch.where(opening, cur_len[ee_s], torch.zeros_like(seg_len))
            inv = torch.empty_like(order); inv[order] = torch.arange(order.numel(), device=device)
            ep_rets = seg_ret[inv].tolist(); ep_lens = seg_len[inv].tolist()
        # carry the open episodes into the next segment
        last_end = torch.where(done, self.step_idx.expand(T, n), torch.zeros((T, n), dtype=torch.int64, device=device)).amax(0)   # 1-based
        tail_ret = csum[-1] - torch.where(last_end > 0, csum[(last_end - 1).clamp(min=0), torch.arange(n, device=device)], torch.zeros_like(csum[-1]))
        self.cur_ret = torch.where(last_end > 0, tail_ret, cur_ret + tail_ret)
        self.cur_len = torch.where(last_end > 0, T - last_end, cur_len + T)
        return ep_rets, ep_lens


def traj_segment_generator(pi, env, horizon, stochastic=True, device=None, first_reset="rsi", fused=False):
    """Batched `traj_segment_generator` (src/trpo.py:27-80): N envs advance in lock step on the device.

    pi: policy.MlpPolicy; env: DPVecEnv created with autoreset="init" — the kernel then applies, on `done`, exactly what
    the reference does on the host (`env.reset(); ob = env.env.reset_model_init()`, :77-79) and returns the fresh
    episode's observation.  Yields, every `horizon` steps, the reference's segment dict with a leading [T, N] shape:
    ob [T,N,56] f32, ac / prevac [T,N,28] f32, rew / vpred [T,N] f32, new [T,N] int32 (new[t] = ob[t] starts an episode),
    nextvpred [N], ep_rets / ep_lens (lists of finished episodes, host numbers — the only host transfer, once per segment).

    Per step the loop issues only the policy forward and ONE env launch: the policy writes its action and value straight
    into row t of the segment buffers, and `dm_batch_step` reads that action row and writes the next observation, the
    reward and the done flag straight into rows t+1 / t / t of theirs (float64, as the C ABI produces them).  With `fused=True`
    (see `can_fuse`) even the policy forward is gone: the env step kernel runs it on the observation it has just produced
    (`dm_batch_step_act`), a step is ONE launch, and with `DM_OPT_PIPELINE` on the batch consecutive steps overlap.  The float32
    segment views, `new`, `prevac` and the episode statistics are derived once per segment with [T, N]-wide ops.
    Nothing leaves the device or the stream."""
    c = SegmentCollector(pi, env, horizon, stochastic, device, first_reset, fused=fused)
    while True:
        c.launch()
        yield c.collect()


def pipelined_segment_generator(pi, envs, horizon, stochastic=True, first_reset="rsi", fused=False):
    """The same segments from SEVERAL env batches (e.g. two halves of a GPU's envs) stepped concurrently, each on its own CUDA
    stream: the whole T-step chain of every batch (policy forward -> env step -> policy forward ...) is enqueued without a host
    wait, so while one batch's env kernel drains its last, cheap workgroups the other batch's policy / env kernels fill the freed
    wave slots — the overlap `DM_OPT_PIPELINE` gives open-loop stepping, for the closed loop.  Yields one segment dict whose env
    axis is the concatenation of the batches (episode lists concatenated in batch order)."""
    import torch
    cols = [SegmentCollector(pi, e, horizon, stochastic, None, first_reset, stream=torch.cuda.Stream(device=pi.device), fused=fused) for e in envs]
    while True:
        if getattr(pi, "_dirty", False) or getattr(pi, "_packed", None) is None:
            pi.pack()                                  # once, on the current stream, before the side streams fork from it
        for c in cols:
            c.launch()
        segs = [c.collect() for c in cols]
        for sg in segs:
            sg.finish_episode_stats()                  # (the lists are concatenated below)
        out = {}
        for k in segs[0]:
            if k in ("ep_rets", "ep_lens"):
                out[k] = [x for sg in segs for x in sg[k]]
            elif k == "nextvpred":
                out[k] = torch.cat([sg[k] for sg in segs], 0)
            else:
                out[k] = torch.cat([sg[k] for sg in segs], 1)
        yield out


def flatten_segment(seg):
    """[T, N, ...] -> [T*N, ...] views env-major (each env's T steps contiguous), the layout the reference's learner
    consumes when several workers' segments are concatenated."""
    out = {}
    for k in ("ob", "ac", "prevac", "rew", "vpred", "new", "adv", "tdlamret"):
        if k in seg:
            v = seg[k]
            out[k] = v.transpose(0, 1).reshape((-1,) + tuple(v.shape[2:]))
    return out


class DoubleBufferedGather:
    """Per-horizon all-gather of the rollout block, overlapped with stepping: two [T, n, 87] blocks per rank; while block k
    travels (async all-gather on the collective's own stream) the envs fill block 1 - k.  `row(t)` returns the row to fill at
    step t (first making sure that this block's previous gather has finished), `commit(t)` launches the gather when step t
    completes a horizon, `drain()` waits for everything outstanding.  Single-process runs degenerate to one local block.

    Backend "nccl" (RCCL) gathers device tensors in place.  Backend "gloo" has no device all-gather: a block on a GPU is first
    copied to pinned host memory (stream-ordered), the gather runs between host buffers (async, gloo's own threads) and the
    result stays on the host in `gathered_host[k]` — the path the world-2 tests and single-GPU multi-rank runs use."""

    def __init__(self, horizon, n_local, device="cpu", world=None, collective=None):
        """collective: run the multi-rank code path (two blocks, the all-gather into `gathered`) — default: when the process group has more
        than one rank; True forces it for a group of ONE rank, which executes the same RCCL calls where only one GPU is visible."""
        import torch
        import torch.distributed as dist
        if world is None:
            world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        self.T, self.n, self.world = int(horizon), int(n_local), int(world)
        self.collective = (self.world > 1) if collective is None else bool(collective)
        nb = 2 if self.collective else 1
        self.blocks = [torch.zeros((self.T, self.n, ROW), dtype=torch.float32, device=device) for _ in range(nb)]
        on_gpu = torch.device(device).type == "cuda"
        self.host_staged = self.collective and on_gpu and dist.get_backend() == "gloo"
        self.gathered = self.gathered_host = self.host_blocks = None
        if self.collective and self.host_staged:
            self.host_blocks = [torch.zeros((self.T, self.n, ROW), dtype=torch.float32).pin_memory() for _ in range(nb)]
            self.gathered_host = [torch.empty((self.world * self.T, self.n, ROW), dtype=torch.float32) for _ in range(nb)]
        elif self.collective:
            self.gathered = [torch.empty((self.world * self.T, self.n, ROW), dtype=torch.float32, device=device) for _ in range(nb)]
        self.pending = [None] * nb
        self.completed = 0                      # gathers launched so far

    def _k(self, t):
        return (t // self.T) % len(self.blocks)

    def row(self, t):
        k = self._k(t)
        if t % self.T == 0 and self.pending[k] is not None:
            self.pending[k].wait(); self.pending[k] = None
        return self.blocks[k][t % self.T]

    def block(self, t):
        """The whole [T, n, 87] block step t belongs to (for producers that fill a block in one go at the end of a horizon)."""
        k = self._k(t)
        if self.pending[k] is not None:
            self.pending[k].wait(); self.pending[k] = None
        return self.blocks[k]

    def commit(self, t):
        """Call after step t's row has been written.  Returns the index of the gathered buffer when a gather was launched."""
        if self.collective and (t + 1) % self.T == 0:
            import torch
            import torch.distributed as dist
            k = self._k(t)
            if self.host_staged:
                self.host_blocks[k].copy_(self.blocks[k], non_blocking=True)
                torch.cuda.current_stream(self.blocks[k].device).synchronize()      # the host copy is complete before gloo reads it
                self.pending[k] = dist.all_gather_into_tensor(self.gathered_host[k], self.host_blocks[k], async_op=True)
            else:
                self.pending[k] = dist.all_gather_into_tensor(self.gathered[k], self.blocks[k], async_op=True)
            self.completed += 1
            return k
        return None

    def result(self, k):
        """The gathered [world * T, n, 87] buffer of slot k (device tensor, or the host tensor on the host-staged path)."""
        return self.gathered_host[k] if self.host_staged else self.gathered[k]

    def drain(self):
        for k in range(len(self.pending)):
            if self.pending[k] is not None:
                self.pending[k].wait(); self.pending[k] = None
