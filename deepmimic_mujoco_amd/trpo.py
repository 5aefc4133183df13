"""TRPO learner of the reference (`src/trpo.py:97-319`, hyper-parameters of `train()` `:338-353`) on PyTorch-ROCm, fed by the
device-resident rollouts of `rollout.traj_segment_generator`.

Same algorithm, line for line in meaning, at batched scale:

    seg      = next(seg_gen);  add_vtarg_and_adv(seg, gamma, lam)                         (:235-236)
    atarg    = (adv - mean) / std                 [per rank, like the reference]           (:240)
    ob_rms.update(ob)                             [all-reduced moments]                    (:242)
    old      = pi                                                                          (:247 assign_old_eq_new)
    losses, g = surrogate + ent bonus, flat gradient w.r.t. pol* + logstd; all-mean        (:249-251)
    stepdir  = CG(F + damping I, g, 10 iterations), F v = Hessian-vector product of mean KL(old || pi) on every 5th sample,
               all-mean per product                                                        (:229, :245, :256; src/cg.py:2-34)
    fullstep = stepdir / sqrt(0.5 stepdir . F stepdir / max_kl);  backtracking line search: <= 10 halvings until
               KL <= 1.5 max_kl and the surrogate improved                                 (:258-283)
    value fn : vf_iters epochs of Adam over shuffled minibatches on (vpred - tdlamret)^2, gradients all-mean'd,
               the MpiAdam update rule of src/mpi_adam.py:21-35; ob_rms updated per minibatch (:288-296)

What replaces what: TF1 graph functions -> torch autograd (flat gradient, double back-prop for F v); `MPI.Allreduce` /
`MpiAdam` -> `torch.distributed.all_reduce` (RCCL over xGMI with the "nccl" backend, gloo in the CPU tests); the host
numpy batch of 256 samples -> the [T, N] device segment (T x N samples per rank; nothing leaves the GPU except the scalars
that are logged).  The value-function minibatch size is a parameter: the reference's 128 is kept as default, batched runs
use a few thousand (a 1M-sample segment at 128 would be 8 000 sequential Adam steps per epoch).
"""
import math
import os
import time
from collections import deque

import torch

from .rollout import add_vtarg_and_adv, flatten_segment, traj_segment_generator, pipelined_segment_generator

POL_KEYS = ("polfc1/w", "polfc1/b", "polfc2/w", "polfc2/b", "polfinal/w", "polfinal/b", "logstd")   # var_list   (:139)
VF_KEYS = ("vffc1/w", "vffc1/b", "vffc2/w", "vffc2/b", "vffinal/w", "vffinal/b")                     # vf_var_list (:140)


def _world(group=None):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group)
    return 1


def allmean(x, group=None, force=False):
    """`allmean` of src/trpo.py:175-180: element-wise mean over ranks (in place; identity for a single process).  force: issue the
    all-reduce even in a group of one rank (executes the collective where only one GPU is visible)."""
    import torch.distributed as dist
    n = _world(group)
    if n > 1 or (force and dist.is_available() and dist.is_initialized()):
        dist.all_reduce(x, op=dist.ReduceOp.SUM, group=group)
        x /= n
    return x


def cg(f_Ax, b, cg_iters=10, residual_tol=1e-10, sync_free=None):
    """Conjugate gradient of src/cg.py:2-34 (Demmel p. 312) on a flat device vector.  On a GPU the `rdotr < residual_tol: break` of :30-31 is
    applied WITHOUT reading the residual back every iteration (ten host round trips per update): once it has fallen below the tolerance the
    remaining iterations leave x untouched, exactly where the break would have."""
    p = b.clone()
    r = b.clone()
    x = torch.zeros_like(b)
    rdotr = r.dot(r)
    on_device = (b.device.type == "cuda") if sync_free is None else bool(sync_free)
    live = torch.ones((), dtype=b.dtype, device=b.device) if on_device else None       # 1 until the residual test would have broken out
    for _ in range(cg_iters):
        z = f_Ax(p)
        v = rdotr / p.dot(z)
        if on_device:
            x = torch.where(live > 0, x + v * p, x)        # a where on x, not a zero step: past convergence p and v may hold 0/0
        else:
            x += v * p
        r -= v * z
        newrdotr = r.dot(r)
        mu = newrdotr / rdotr
        p = r + mu * p
        rdotr = newrdotr
        if on_device:
            live = live * (rdotr >= residual_tol).to(b.dtype)
        elif float(rdotr) < residual_tol:
            break
    return x


def flat(tensors):
    return torch.cat([t.reshape(-1) for t in tensors])


def explained_variance(ypred, y):
    """src/utils/math_util.py:25-38."""
    vary = torch.var(y, unbiased=False)
    return float("nan") if float(vary) == 0 else float(1 - torch.var(y - ypred, unbiased=False) / vary)


class MpiAdam:
    """src/mpi_adam.py:7-50 on device tensors: gradients are all-mean'd, then the bias-corrected Adam step."""

    def __init__(self, params, beta1=0.9, beta2=0.999, epsilon=1e-8, group=None):
        self.params = list(params)
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.beta1, self.beta2, self.epsilon, self.group = beta1, beta2, epsilon, group
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        self.t = 0

    def getflat(self):
        return flat([p.detach() for p in self.params])

    def setfromflat(self, theta):
        o = 0
        with torch.no_grad():
            for p in self.params:
                n = p.numel()
                p.copy_(theta[o:o + n].reshape(p.shape))
                o += n

    def update(self, localg, stepsize):
        if self.t % 100 == 0:
            self.check_synced()
        g = allmean(localg.to(torch.float32).clone(), self.group)
        self.t += 1
        a = stepsize * math.sqrt(1 - self.beta2 ** self.t) / (1 - self.beta1 ** self.t)
        self.m.mul_(self.beta1).add_(g, alpha=1 - self.beta1)
        self.v.mul_(self.beta2).addcmul_(g, g, value=1 - self.beta2)
        step = (-a) * self.m / (torch.sqrt(self.v) + self.epsilon)
        self.setfromflat(self.getflat() + step)

    def sync(self):
        import torch.distributed as dist
        if _world(self.group) > 1:
            theta = self.getflat().clone()
            dist.broadcast(theta, src=0, group=self.group)
            self.setfromflat(theta)

    def check_synced(self):
        import torch.distributed as dist
        if _world(self.group) > 1:
            theta = self.getflat().clone()
            root = theta.clone()
            dist.broadcast(root, src=0, group=self.group)
            if not bool((root == theta).all()):
                raise AssertionError("value-function parameters diverged across ranks")


class _VfGraph:
    """One value-fit minibatch step (TrpoLearner._vf_step) captured as a hipGraph.  The epoch's shuffled samples sit in static
    [nb, bs, .] buffers; a device-side counter picks the minibatch and the step's Adam scale (precomputed on the host for the
    whole epoch), so a replay needs no host-side work at all."""

    def __init__(self, learner, ob, ret, bs):
        self.ok = False
        self.L = learner
        pi, ad = learner.pi, learner.vfadam
        n = ob.shape[0]
        self.bs, self.nb, self.n = bs, n // bs, n
        dev = ob.device
        self.ob_s = torch.zeros((self.nb, bs) + tuple(ob.shape[1:]), dtype=ob.dtype, device=dev)
        self.ret_s = torch.zeros((self.nb, bs), dtype=ret.dtype, device=dev)
        self.a_s = torch.zeros(self.nb, dtype=torch.float32, device=dev)
        self.ctr = torch.zeros(1, dtype=torch.long, device=dev)
        self.one = torch.ones(1, dtype=torch.long, device=dev)
        self.cnt_add = torch.tensor(float(bs), dtype=torch.float64, device=dev)
        rms = pi.ob_rms
        # snapshot: warm-up and capture run the step for real
        saved = [t.clone() for t in (rms.sum, rms.sumsq, rms.count, ad.m, ad.v)] + [p.detach().clone() for p in ad.params]
        # Whatever happens in warm-up or capture (an exception sends the learner to its eager loop), the parameters, the Adam moments and
        # the observation filter go back to the snapshot: the fallback must not train on state the throw-away steps have moved.
        try:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(3):
                    self.ctr.zero_()              # always minibatch 0: the warm-up must not index past nb (nb may be < 3)
                    self._body()
            torch.cuda.current_stream(dev).wait_stream(side)
            self.ctr.zero_()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self._body()
            self.ok = True
        finally:
            torch.cuda.synchronize(dev)
            with torch.no_grad():
                for t, sv in zip([rms.sum, rms.sumsq, rms.count, ad.m, ad.v] + list(ad.params), saved):
                    t.copy_(sv)
                rms._refresh()
            self.ctr.zero_()

    def matches(self, ob, ret, bs):
        return ob.shape[0] == self.n and bs == self.bs and ob.dtype == self.ob_s.dtype and ob.device == self.ob_s.device

    def _body(self):
        L = self.L
        pi, ad, rms = L.pi, L.vfadam, L.pi.ob_rms
        mbob = torch.index_select(self.ob_s, 0, self.ctr)[0]
        mbret = torch.index_select(self.ret_s, 0, self.ctr)[0]
        a = torch.index_select(self.a_s, 0, self.ctr)[0]
        # RunningMeanStd.update, device-only
        x = mbob.to(torch.float64)
        with torch.no_grad():
            rms.sum += x.sum(0).reshape(rms.shape)
            rms.sumsq += (x * x).sum(0).reshape(rms.shape)
            rms.count += self.cnt_add
            rms._refresh()
        vpred = pi.forward_value(mbob)
        vferr = ((vpred - mbret) ** 2).mean()
        g = flat(torch.autograd.grad(vferr, L.vf)).to(torch.float32)
        with torch.no_grad():                                              # MpiAdam.update with the step scale from the table
            ad.m.mul_(ad.beta1).add_(g, alpha=1 - ad.beta1)
            ad.v.mul_(ad.beta2).addcmul_(g, g, value=1 - ad.beta2)
            step = (-a) * ad.m / (torch.sqrt(ad.v) + ad.epsilon)
            ad.setfromflat(ad.getflat() + step)
            self.ctr += self.one

    def run_epoch(self, ob, ret, inds):
        L, ad = self.L, self.L.vfadam
        used = inds[:self.nb * self.bs]
        self.ob_s.view(self.nb * self.bs, *ob.shape[1:]).copy_(ob[used])
        self.ret_s.view(-1).copy_(ret[used])
        a = []
        for k in range(self.nb):
            t = ad.t + 1 + k
            a.append(L.vf_stepsize * math.sqrt(1 - ad.beta2 ** t) / (1 - ad.beta1 ** t))
        self.a_s.copy_(torch.tensor(a, dtype=torch.float32), non_blocking=False)
        self.ctr.zero_()
        for _ in range(self.nb):
            self.graph.replay()
        ad.t += self.nb


class TrpoLearner:
    """One policy/value update per segment; the reference's `learn()` body between `seg_gen.__next__()` and the logging."""

    def __init__(self, pi, *, max_kl=0.01, cg_iters=10, cg_damping=0.1, gamma=0.995, lam=0.97, entcoeff=0.0,
                 vf_iters=3, vf_stepsize=1e-3, vf_batch_size=128, fvp_subsample=5, group=None, seed=0, vf_graph=None, vf_native=None, pg_native=None):
        self.pi = pi
        self.max_kl, self.cg_iters, self.cg_damping = max_kl, cg_iters, cg_damping
        self.gamma, self.lam, self.entcoeff = gamma, lam, entcoeff
        self.vf_iters, self.vf_stepsize, self.vf_batch_size = vf_iters, vf_stepsize, vf_batch_size
        self.fvp_subsample = fvp_subsample
        # value-fit minibatch steps as one captured hipGraph each (single-process GPU runs; None = when possible)
        self.vf_graph = vf_graph
        import os
        self.vf_overlap = os.environ.get("DM_VF_OVERLAP", "1") != "0"     # value fit on a second stream beside the policy step (both on kernels)
        self.vf_share = os.environ.get("DM_VF_SHARE", "1") != "0"         # ... and the policy launches leave it its CUs while it runs (_pg_share_begin)
        self._share = None
        self._vf_stream = None
        self._rms_pol = None
        self.vf_epoch_filter = os.environ.get("DM_VF_EPOCH_FILTER", "1") != "0"   # obs-filter sums of an epoch's minibatches up front (False: per minibatch)
        self._vfg = None
        # ... or as the hand-written kernels of csrc/vf_kernel.h (three launches per minibatch, one C call per epoch; None = when possible)
        self.vf_native = vf_native
        self._vf_scratch = None
        # the policy half (losses + flat gradient, Fisher-vector products, line-search losses) as the kernels of csrc/pg_kernel.h
        # (None = when possible: CUDA tensors, the reference's 56-100-100-28 policy; multi-rank runs all-mean the kernels' results)
        self.pg_native = pg_native
        self._pg_scratch = None
        self.group = group
        for k in POL_KEYS + VF_KEYS:
            pi.params[k].requires_grad_(True)
        self.pol = [pi.params[k] for k in POL_KEYS]
        self.vf = [pi.params[k] for k in VF_KEYS]
        self.vfadam = MpiAdam(self.vf, group=group)
        self._perm_gen = torch.Generator(device=pi.device)
        self._perm_gen.manual_seed(int(seed))
        self.perm_source = None        # tests: callable(n) -> index tensor replacing the shuffles of `dataset.iterbatches` (:289)
        self._perms = []               # shuffles drawn ahead of the update that uses them (`_prefetch_perms`)
        self.last = {}                 # flat g / stepdir / fullstep of the last update (diagnostics, parity tests)
        self.sync_from_root()

    # ---- flat parameter access (U.GetFlat / U.SetFromFlat, :144-145) ----------------------------------------------------
    def get_flat(self):
        return flat([p.detach() for p in self.pol]).clone()

    def set_from_flat(self, theta):
        o = 0
        with torch.no_grad():
            for p in self.pol:
                n = p.numel()
                p.copy_(theta[o:o + n].reshape(p.shape))
                o += n

    def sync_from_root(self):
        """:182-186 — rank 0's initial parameters everywhere."""
        import torch.distributed as dist
        if _world(self.group) > 1:
            th = self.get_flat()
            dist.broadcast(th, src=0, group=self.group)
            self.set_from_flat(th)
            self.vfadam.sync()

    # ---- losses (:118-134) -------------------------------------------------------------------------------------------------
    def _pd(self, ob):
        return self.pi.forward_mean(ob), self.pi.params["logstd"]

    @staticmethod
    def _kl(mean0, logstd0, mean1, logstd1):
        """DiagGaussianPd.kl(self = 0, other = 1), src/distributions.py:235-237."""
        return (logstd1 - logstd0 + (torch.exp(2 * logstd0) + (mean0 - mean1) ** 2) / (2.0 * torch.exp(2 * logstd1)) - 0.5).sum(-1)

    @staticmethod
    def _neglogp(x, mean, logstd):
        return 0.5 * (((x - mean) / torch.exp(logstd)) ** 2).sum(-1) + 0.5 * math.log(2.0 * math.pi) * x.shape[-1] + logstd.sum(-1)

    def _losses(self, ob, ac, atarg, old_mean, old_logstd):
        mean, logstd = self._pd(ob)
        meankl = self._kl(old_mean, old_logstd, mean, logstd).mean()
        meanent = (logstd + 0.5 * math.log(2.0 * math.pi * math.e)).sum(-1).mean()
        entbonus = self.entcoeff * meanent
        ratio = torch.exp(self._neglogp(ac, old_mean, old_logstd) - self._neglogp(ac, mean, logstd))     # pnew / pold
        surrgain = (ratio * atarg).mean()
        optimgain = surrgain + entbonus
        return optimgain, meankl, entbonus, surrgain, meanent

    loss_names = ("optimgain", "meankl", "entloss", "surrgain", "entropy")

    # ---- value fit (:288-296) --------------------------------------------------------------------------------------------
    def _vf_step(self, mbob, mbret):
        """One minibatch of the value fit: obs-filter update (:293), squared-error gradient, MpiAdam step."""
        self.pi.ob_rms.update(mbob, group=self.group)
        vpred = self.pi.forward_value(mbob)
        vferr = ((vpred - mbret) ** 2).mean()
        gv = flat(torch.autograd.grad(vferr, self.vf))
        self.vfadam.update(gv, self.vf_stepsize)

    def _vf_native_ready(self, ob, ret):
        """The value fit as csrc/vf_kernel.h's kernels (dm_vf_fit_epoch): single-process GPU runs with the reference's 56-100-100-1 value
        net.  Same arithmetic as `_vf_step` in float32 (obs-filter sums in float64), sums in a fixed order."""
        if self.vf_native is False or ob.device.type != "cuda" or _world(self.group) > 1:
            return False
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return False
        p = self.pi.params
        ok = (ob.dtype == torch.float32 and ob.dim() == 2 and ob.shape[1] == 56 and ret.dtype == torch.float32 and getattr(self.pi, "native", False)
              and tuple(p["vffc1/w"].shape) == (56, 100) and tuple(p["vffc2/w"].shape) == (100, 100) and tuple(p["vffinal/w"].shape) == (100, 1)
              and all(p[k].dtype == torch.float32 for k in VF_KEYS) and tuple(self.pi.ob_rms.shape) == (56,))
        if not ok and self.vf_native is True:
            raise ValueError("the native value fit needs float32 [n, 56] observations and the 56-100-100-1 value net on a GPU")
        return ok

    def _vf_native_epoch(self, ob, ret, inds, bs):
        import ctypes as C
        from . import _abi as A
        L = A.load()
        ad, rms = self.vfadam, self.pi.ob_rms
        n = ob.shape[0]
        nb = n // bs                                                        # include_final_partial_batch=False
        used = inds[:nb * bs]
        ob_s = ob[used].contiguous(); ret_s = ret[used].contiguous()
        theta = ad.getflat().to(torch.float32).contiguous()
        assert theta.numel() == L.dm_vf_param_count()
        need = int(L.dm_vf_scratch_bytes(int(nb), int(bs)))
        if self._vf_scratch is None or self._vf_scratch.numel() < need or self._vf_scratch.device != ob.device:
            self._vf_scratch = torch.empty(need, dtype=torch.uint8, device=ob.device)
        scale = (C.c_float * nb)(*[self.vf_stepsize * math.sqrt(1 - ad.beta2 ** (ad.t + 1 + k)) / (1 - ad.beta1 ** (ad.t + 1 + k)) for k in range(nb)])
        pp = lambda t: C.c_void_p(t.data_ptr())
        A.check(L.dm_vf_fit_epoch(pp(ob_s), pp(ret_s), nb, int(bs), pp(theta), pp(ad.m), pp(ad.v), scale, float(ad.beta1), float(ad.beta2), float(ad.epsilon),
                                  pp(rms.sum), pp(rms.sumsq), pp(rms.count), pp(rms.mean), pp(rms.std), pp(self._vf_scratch),
                                  C.c_void_p(torch.cuda.current_stream(ob.device).cuda_stream), 1 if self.vf_epoch_filter else 0), L)
        ad.setfromflat(theta)
        ad.t += nb

    def _vf_graph_ready(self, ob, ret, bs):
        """The minibatch step is ~60 launches of tiny kernels (CPU-bound: 0.7 ms each, 384 of them per update at 4 096 envs x 128
        steps).  On a single-process GPU run it is captured once as a hipGraph and replayed; multi-rank runs (collectives inside the
        step) and CPU runs keep the eager loop.  Same operations in the same order: results are identical."""
        if self.vf_graph is False or ob.device.type != "cuda" or _world(self.group) > 1:
            return False
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return False
        g = self._vfg
        if g is not None and g.ok and g.matches(ob, ret, bs):
            return True
        if g is not None and not g.ok:
            return False
        try:
            self._vfg = _VfGraph(self, ob, ret, bs)
        except Exception as ex:                                            # capture not possible here: stay eager, say so once
            import warnings
            warnings.warn("value-fit graph capture failed (%s: %s); using the eager loop" % (type(ex).__name__, ex))
            self._vfg = _VfGraph.__new__(_VfGraph); self._vfg.ok = False
            if self.vf_graph is True:
                raise
            return False
        return True

    # ---- policy half as hand-written kernels (csrc/pg_kernel.h) -------------------------------------------------------------
    def _pg_native_ready(self, ob, ac):
        if self.pg_native is False or ob.device.type != "cuda":
            return False
        p = self.pi.params
        ok = (ob.dtype == torch.float32 and ob.dim() == 2 and ob.shape[1] == 56 and ac.dtype == torch.float32 and ac.dim() == 2 and ac.shape[1] == 28
              and getattr(self.pi, "native", False) and tuple(p["polfc1/w"].shape) == (56, 100) and tuple(p["polfc2/w"].shape) == (100, 100)
              and tuple(p["polfinal/w"].shape) == (100, 28) and p["logstd"].numel() == 28 and all(p[k].dtype == torch.float32 for k in POL_KEYS)
              and tuple(self.pi.ob_rms.shape) == (56,))
        if not ok and self.pg_native is True:
            raise ValueError("the native policy update needs float32 [n, 56] observations / [n, 28] actions and the 56-100-100-28 policy on a GPU")
        return ok

    def _pg_call(self, name, *args):
        import ctypes as C
        from . import _abi as A
        L = A.load()
        A.check(getattr(L, name)(*[C.c_void_p(a.data_ptr()) if torch.is_tensor(a) else a for a in args]), L)

    def _pg_buffers(self, dev):
        from . import _abi as A
        L = A.load()
        if self._pg_scratch is None or self._pg_scratch.device != dev:
            assert L.dm_pg_param_count() == sum(p.numel() for p in self.pol)
            self._pg_scratch = torch.empty(int(L.dm_pg_scratch_bytes()), dtype=torch.uint8, device=dev)
        return self._pg_scratch

    # ---- CU sharing between the policy step and the value fit running beside it ----
    # Both halves' kernels fill a CU's LDS with one workgroup, so a full-grid policy launch (256 blocks) and the fit's gradient kernel never
    # run side by side: the fit's launches would only squeeze in between policy launches and the two "overlapped" halves take the sum of
    # their times.  While the fit is expected to be running, the policy launches therefore take only the CUs the fit's grid leaves; once
    # it should be through they take all of them.  WHICH launches are the narrow ones is decided from a cost model of the launch sequence
    # (ns per sample on a full MI355X, measured: profiles/r04_train_kernels.md), not from the clock, so that a seeded run reproduces
    # bit for bit (the gradient sums are taken in block order: a function of the grid size).
    PG_GRAD_NS, PG_FVP_NS, PG_LOSS_NS = 2.06, 2.53, 0.89
    VF_STEP_US, VF_EPOCH_US = 26.5, 170.0
    N_CU = 256                  # the chip the cost constants above were measured on (MI355X)

    def _cu_count(self, dev):
        """Compute units of the device the update runs on; the sharing plan is only valid where it equals N_CU (the constants are MI355X measurements and
        'the CUs the fit leaves' means nothing on another chip): elsewhere sharing is off and every policy launch takes its full grid."""
        c = getattr(self, "_cus", None)
        if c is None or c[0] != dev:
            try:
                c = (dev, int(torch.cuda.get_device_properties(dev).multi_processor_count))
            except Exception:       # noqa: BLE001 — unknown device: no sharing
                c = (dev, 0)
            self._cus = c
        return c[1]

    def _pg_share_begin(self, n, bs, dev=None):
        nb = n // bs
        vf_blocks = (bs + 31) // 32
        if dev is not None and self._cu_count(dev) != self.N_CU:
            self._share = None                                              # not the chip the plan was measured on
            return
        if vf_blocks >= self.N_CU * 3 // 4:
            self._share = None                                              # the fit wants (nearly) the whole chip anyway: nothing to leave
            return
        vf_ms = self.vf_iters * (nb * self.VF_STEP_US * max(1.0, vf_blocks / 128.0) + self.VF_EPOCH_US) * 1e-3
        self._share = {"t": 0.0, "until": vf_ms, "blocks": self.N_CU - vf_blocks}

    def _pg_grid(self, cost_ns):
        sh = getattr(self, "_share", None)
        if sh is None:
            return 0
        ms = cost_ns * 1e-6
        if sh["t"] >= sh["until"]:
            sh["t"] += ms
            return 0
        sh["t"] += ms * self.N_CU / sh["blocks"]
        return int(sh["blocks"])

    def _pg_losses(self, ob, ac, atarg, old_mean, old_logstd, theta, write_old, with_grad):
        """-> (losses [5] float32 like `_losses`: optimgain, meankl, entbonus, surrgain, meanent; flat gradient or None)"""
        import ctypes as C
        dev = ob.device
        sc = self._pg_buffers(dev)
        rms_mean, rms_std = self._rms_pol or (self.pi.ob_rms.mean, self.pi.ob_rms.std)
        out = torch.empty(2, dtype=torch.float64, device=dev)
        g = torch.empty(theta.numel(), dtype=torch.float32, device=dev) if with_grad else None
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        self._pg_call("dm_pg_losses", ob, int(ob.shape[0]), ac, atarg, old_mean, old_logstd, 1 if write_old else 0, theta, rms_mean, rms_std,
                      C.c_double(float(self.entcoeff)), 1 if with_grad else 0, g if with_grad else C.c_void_p(0), out, sc, st,
                      self._pg_grid((self.PG_GRAD_NS if with_grad else self.PG_LOSS_NS) * ob.shape[0]))
        logstd = theta[-28:]
        meanent = (logstd + 0.5 * math.log(2.0 * math.pi * math.e)).sum()
        surr, kl = out[0].to(torch.float32), out[1].to(torch.float32)
        entbonus = self.entcoeff * meanent
        return torch.stack([surr + entbonus, kl, entbonus, surr, meanent]), g

    def _pg_fvp(self, ob, theta, v):
        import ctypes as C
        dev = ob.device
        sc = self._pg_buffers(dev)
        rms_mean, rms_std = self._rms_pol or (self.pi.ob_rms.mean, self.pi.ob_rms.std)
        k = int(self.fvp_subsample)
        nf = (int(ob.shape[0]) + k - 1) // k                                # rows of ob[::k]
        hv = torch.empty_like(v)
        self._pg_call("dm_pg_fvp", ob, k, nf, theta, v.contiguous(), rms_mean, rms_std, hv, sc, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream),
                      self._pg_grid(self.PG_FVP_NS * nf))
        return hv

    def _rms_update(self, ob):
        """`pi.ob_rms.update(ob)` (:242): one launch (dm_rms_update) for a float32 batch on the GPU of a single-process run, else the tensor ops."""
        rms = self.pi.ob_rms
        if (ob.device.type == "cuda" and ob.dtype == torch.float32 and ob.dim() == 2 and ob.shape[1] == 56 and ob.is_contiguous() and _world(self.group) == 1
                and tuple(rms.shape) == (56,) and torch.is_tensor(rms.count) and rms.sum.is_cuda and getattr(self.pi, "native", False)):
            import ctypes as C
            from . import _abi as A
            L = A.load()
            if getattr(self, "_rms_scratch", None) is None or self._rms_scratch.device != ob.device:
                self._rms_scratch = torch.empty(int(L.dm_rms_scratch_bytes()), dtype=torch.uint8, device=ob.device)
            p = lambda x: C.c_void_p(x.data_ptr())
            A.check(L.dm_rms_update(p(ob), int(ob.shape[0]), p(rms.sum), p(rms.sumsq), p(rms.count), p(rms.mean), p(rms.std), p(self._rms_scratch),
                                    C.c_void_p(torch.cuda.current_stream(ob.device).cuda_stream)), L)
            return
        rms.update(ob, group=self.group)

    def _next_perm(self, n, dev):
        """The next shuffle of `dataset.iterbatches` (:289) from this learner's generator — taken from the ones drawn ahead when they fit."""
        while self._perms:
            p = self._perms.pop(0)
            if p.numel() == n and p.device == dev:
                if self._vf_stream is not None:
                    torch.cuda.current_stream(dev).wait_stream(self._vf_stream)   # (drawn on the fit's stream; a no-op when that is the consumer)
                return p
            self._perms = []           # the segment size changed: the generator's stream goes on from here (nothing drawn is re-used)
        return torch.randperm(n, device=dev, generator=self._perm_gen)

    def _prefetch_perms(self, n, dev):
        """A shuffle of half a million indices is a 19-pass radix sort (0.65 ms): the next update's three are drawn on the fit's stream once this
        update no longer waits for it — they run beside the rollout's bookkeeping instead of at the head of every epoch of the fit.  Same
        generator, same order of draws: the shuffles are the ones an update drawing them itself would get."""
        if self.perm_source is None and not self._perms:
            self._perms = [torch.randperm(n, device=dev, generator=self._perm_gen) for _ in range(self.vf_iters)]

    # ---- one update ----------------------------------------------------------------------------------------------------------
    def update(self, seg):
        pi = self.pi
        import os
        prof = {} if os.environ.get("DM_TRPO_PROFILE") else None             # phase times (ms, device-synchronised) in stats["profile_ms"]
        t_last = [time.perf_counter()]

        def tick(name):
            if prof is None:
                return
            if seg["ob"].device.type == "cuda":
                torch.cuda.current_stream(seg["ob"].device).synchronize()   # (this stream only: the value fit may be running beside it; its own entry is then what is left to wait for)
            now = time.perf_counter()
            prof[name] = prof.get(name, 0.0) + (now - t_last[0]) * 1e3
            t_last[0] = now
        tick("_start")
        add_vtarg_and_adv(seg, self.gamma, self.lam)
        fl = flatten_segment(seg)
        ob, ac, atarg, tdlamret, vpredbefore = fl["ob"], fl["ac"], fl["adv"], fl["tdlamret"], fl["vpred"]
        self._rms_update(ob)                                                # :242 (before :240 here: the value fit's stream forks right after it)
        native_pg = self._pg_native_ready(ob, ac)
        # ---- value function (:288-296).  It shares nothing with the policy step but the obs filter, which it moves on minibatch by minibatch:
        # with both halves on kernels the fit is enqueued NOW on a second stream and runs beside the policy step (its gradient kernel holds one
        # block per CU on half the CUs); the policy step reads a copy of the filter as :242 left it — the reference's order of effects.
        n = ob.shape[0]
        bs = min(self.vf_batch_size, n)
        vf_native = self._vf_native_ready(ob, tdlamret)
        graphed = (not vf_native) and self._vf_graph_ready(ob, tdlamret, bs)

        def fit_value():
            for _ in range(self.vf_iters):
                inds = self.perm_source(n).to(ob.device) if self.perm_source is not None else self._next_perm(n, ob.device)
                if vf_native:
                    self._vf_native_epoch(ob, tdlamret, inds, bs)
                    continue
                if graphed:
                    self._vfg.run_epoch(ob, tdlamret, inds)
                    continue
                for o in range(0, n - bs + 1, bs):                          # include_final_partial_batch=False
                    mb = inds[o:o + bs]
                    self._vf_step(ob[mb], tdlamret[mb])

        overlap = bool(native_pg and vf_native and self.vf_overlap)
        rms_pol = (pi.ob_rms.mean, pi.ob_rms.std)
        if overlap:
            rms_pol = (pi.ob_rms.mean.clone(), pi.ob_rms.std.clone())
            if self._vf_stream is None or self._vf_stream.device != ob.device:
                self._vf_stream = torch.cuda.Stream(device=ob.device)
            main = torch.cuda.current_stream(ob.device)
            self._vf_stream.wait_stream(main)
            with torch.cuda.stream(self._vf_stream):
                fit_value()
            if self.vf_share:
                self._pg_share_begin(n, bs, ob.device)
        self._rms_pol = rms_pol
        atarg = (atarg - atarg.mean()) / atarg.std(unbiased=False)          # :240
        if native_pg:
            ob = ob.contiguous(); ac = ac.contiguous(); atarg = atarg.to(torch.float32).contiguous()
            theta0 = self.get_flat().contiguous()
            old_logstd = pi.params["logstd"].detach().reshape(-1).clone()
            old_mean = torch.empty((ob.shape[0], 28), dtype=torch.float32, device=ob.device)
            # one launch: oldpi <- pi (:247: old_mean is written), the losses at pi == oldpi and the flat gradient of optimgain (:249)
            lossbefore, g = self._pg_losses(ob, ac, atarg, old_mean, old_logstd, theta0, write_old=True, with_grad=True)
            lossbefore = allmean(lossbefore, self.group)
        else:
            with torch.no_grad():                                           # :247 oldpi <- pi
                old_mean = pi.forward_mean(ob)
                old_logstd = pi.params["logstd"].detach().clone()
            sub = slice(None, None, self.fvp_subsample)                     # :245 fvpargs = [arr[::5] ...]
            ob_f, om_f = ob[sub], old_mean[sub]
            losses = self._losses(ob, ac, atarg, old_mean, old_logstd)
            g = flat(torch.autograd.grad(losses[0], self.pol))
            lossbefore = allmean(torch.stack([l.detach() for l in losses]), self.group)
        g = allmean(g, self.group)
        if hasattr(seg, "finish_episode_stats"):
            seg.finish_episode_stats()                                      # host work (sorting the segment's episode records) while the gradient kernel runs
        tick("gae_filter_losses_grad")
        stats = {}
        if bool(torch.allclose(g, torch.zeros_like(g))):
            stats["note"] = "zero gradient, not updating"
            meanlosses = lossbefore
        else:
            if native_pg:
                # Fisher-vector products: forward-mode J v, then J^T (J v / sigma^2) / N — the exact Hessian of the mean KL at pi == oldpi
                def fisher_vector_product(p):
                    return allmean(self._pg_fvp(ob, theta0, p), self.group) + self.cg_damping * p        # :229
                klgrads = None
            else:
                # Fisher-vector products: gradient of (grad KL . v), the KL graph is built once and re-used by every product
                mean_f, logstd_f = self._pd(ob_f)
                kl_f = self._kl(om_f, old_logstd, mean_f, logstd_f).mean()
                klgrads = flat(torch.autograd.grad(kl_f, self.pol, create_graph=True))

                def fisher_vector_product(p):
                    hv = flat(torch.autograd.grad(klgrads.dot(p), self.pol, retain_graph=True))
                    return allmean(hv, self.group) + self.cg_damping * p    # :229

            stepdir = cg(fisher_vector_product, g, cg_iters=self.cg_iters)
            assert bool(torch.isfinite(stepdir).all())
            shs = 0.5 * stepdir.dot(fisher_vector_product(stepdir))
            tick("cg_fisher_products")
            lm = torch.sqrt(shs / self.max_kl)
            fullstep = stepdir / lm
            expectedimprove = float(g.dot(fullstep))
            self.last = {"g": g.detach().clone(), "stepdir": stepdir.detach().clone(), "fullstep": fullstep.detach().clone(),
                         "shs": float(shs), "lm": float(lm)}
            del klgrads
            surrbefore = float(lossbefore[0])
            stepsize = 1.0
            thbefore = self.get_flat()
            ok = False
            for _ in range(10):                                             # :266-283
                self.set_from_flat(thbefore + fullstep * stepsize)
                if native_pg:
                    meanlosses = allmean(self._pg_losses(ob, ac, atarg, old_mean, old_logstd, self.get_flat().contiguous(), write_old=False, with_grad=False)[0], self.group)
                else:
                    with torch.no_grad():
                        meanlosses = allmean(torch.stack(self._losses(ob, ac, atarg, old_mean, old_logstd)), self.group)
                surr, kl = float(meanlosses[0]), float(meanlosses[1])
                improve = surr - surrbefore
                if not bool(torch.isfinite(meanlosses).all()):
                    pass                                                    # "Got non-finite value of losses -- bad!"
                elif kl > self.max_kl * 1.5:
                    pass                                                    # "violated KL constraint. shrinking step."
                elif improve < 0:
                    pass                                                    # "surrogate didn't improve. shrinking step."
                else:
                    ok = True                                               # "Stepsize OK!"
                    break
                stepsize *= 0.5
            if not ok:
                self.set_from_flat(thbefore)                                # "couldn't compute a good step"
            stats.update(expectedimprove=expectedimprove, improve=improve, stepsize=stepsize if ok else 0.0)
            tick("line_search")

        self._share = None
        if overlap:
            torch.cuda.current_stream(ob.device).wait_stream(self._vf_stream)   # the fit's parameters / filter state before anything after this update
            with torch.cuda.stream(self._vf_stream):
                self._prefetch_perms(n, ob.device)
        else:
            fit_value()
        tick("value_fit")
        if prof is not None:
            prof.pop("_start", None)
            stats["profile_ms"] = {k: round(v, 3) for k, v in prof.items()}
        self._rms_pol = None
        pi.mark_dirty()                                                     # parameters / obs filter changed in place: the native act() repacks
        for name, val in zip(self.loss_names, meanlosses.tolist()):
            stats[name] = val
        stats["ev_tdlam_before"] = explained_variance(vpredbefore, tdlamret)
        return stats


def learn(env, pi, *, timesteps_per_batch=256, max_iters=0, max_timesteps=0, max_seconds=0, callback=None, log=print,
          group=None, log_dir=None, fused=None, **learner_kwargs):
    """`learn()` of src/trpo.py:97-319 over a DPVecEnv (autoreset="init"; or a list of them: pipelined rollouts) and an MlpPolicy.  Stops after `max_iters`
    iterations, `max_timesteps` env steps (global) or `max_seconds`.  Returns the list of per-iteration stat dicts, with
    the reference's log keys (EpLenMean / EpRewMean over the last 40 episodes, EpThisIter, EpisodesSoFar, TimestepsSoFar,
    TimeElapsed, entropy, meankl, optimgain, surrgain, ev_tdlam_before).  With `log_dir`, rank 0 also writes the reference's
    files there: `progress.csv` (logger CSV, src/logger.py:101-135) and `monitor.json.monitor.csv` (bench.Monitor, one row per
    finished episode of rank 0's envs) — readable by the reference's plot_curve.py / load_results."""
    import torch.distributed as dist
    assert sum([max_iters > 0, max_timesteps > 0, max_seconds > 0]) >= 1
    learner = TrpoLearner(pi, group=group, **learner_kwargs)
    if isinstance(env, (list, tuple)):          # several env batches of this rank, stepped concurrently on their own streams
        seg_gen = pipelined_segment_generator(pi, list(env), timesteps_per_batch, stochastic=True)
        n_envs_local = sum(e.num_envs for e in env)
    else:
        # fused (default when possible): the policy step runs inside the env step kernel, one launch per rollout step
        from .rollout import can_fuse
        use_fused = can_fuse(pi, env) if fused is None else bool(fused)
        seg_gen = traj_segment_generator(pi, env, timesteps_per_batch, stochastic=True, fused=use_fused)
        n_envs_local = env.num_envs
    world = _world(group)
    rank = dist.get_rank(group) if world > 1 else 0
    episodes_so_far = timesteps_so_far = iters_so_far = 0
    tstart = time.time()
    lenbuffer, rewbuffer = deque(maxlen=40), deque(maxlen=40)
    history = []
    progress = monitor = None
    if log_dir and rank == 0:
        from .logio import ProgressCsv, MonitorWriter
        os.makedirs(log_dir, exist_ok=True)
        progress = ProgressCsv(os.path.join(log_dir, "progress.csv"))
        monitor = MonitorWriter(os.path.join(log_dir, "monitor.json"), t_start=tstart)
    while True:
        if callback:
            callback(locals(), globals())
        if max_timesteps and timesteps_so_far >= max_timesteps:
            break
        if max_iters and iters_so_far >= max_iters:
            break
        if max_seconds:
            # the deadline is a per-process wall clock: decide collectively (MAX over ranks), or a rank that breaks first leaves
            # the others waiting forever in the next update's all-reduces
            stop = time.time() - tstart >= max_seconds
            if world > 1:
                flag = torch.tensor([1.0 if stop else 0.0], dtype=torch.float32, device=pi.device)
                dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
                stop = bool(flag.item() > 0)
            if stop:
                break
        if os.environ.get("DM_TRPO_PROFILE") and pi.device.type == "cuda":
            torch.cuda.synchronize(pi.device); t_seg = time.perf_counter()
            seg = next(seg_gen)
            torch.cuda.synchronize(pi.device); t_seg = (time.perf_counter() - t_seg) * 1e3
        else:
            seg = next(seg_gen); t_seg = None
        stats = learner.update(seg)
        if t_seg is not None and "profile_ms" in stats:
            stats["profile_ms"]["rollout_segment"] = round(t_seg, 3)
            if "collect_ms" in seg:
                stats["profile_ms"]["of_which_segment_bookkeeping"] = round(float(seg["collect_ms"]), 3)
        if getattr(seg, "info", None):
            stats["rollout"] = dict(seg.info)
        lens, rets = seg["ep_lens"], seg["ep_rets"]
        n_eps = torch.tensor([len(lens), sum(lens), sum(rets)], dtype=torch.float64, device=pi.device)
        if world > 1:                                            # :300-302 allgather of (ep_lens, ep_rets): the sums suffice here
            dist.all_reduce(n_eps, group=group)
        lenbuffer.extend(lens[-40:]); rewbuffer.extend(rets[-40:])
        episodes_so_far += int(n_eps[0]); timesteps_so_far += timesteps_per_batch * n_envs_local * world
        iters_so_far += 1
        stats.update(EpLenMean=float(sum(lenbuffer) / max(1, len(lenbuffer))), EpRewMean=float(sum(rewbuffer) / max(1, len(rewbuffer))),
                     EpLenMeanIter=float(n_eps[1] / max(1.0, float(n_eps[0]))), EpThisIter=int(n_eps[0]), EpisodesSoFar=episodes_so_far,
                     TimestepsSoFar=timesteps_so_far, TimeElapsed=time.time() - tstart, iteration=iters_so_far)
        history.append(stats)
        if progress is not None:
            progress.writekvs({k: stats.get(k) for k in ("EpRewMean", "EpThisIter", "TimestepsSoFar", "EpisodesSoFar", "surrgain", "optimgain",
                                                        "TimeElapsed", "meankl", "entloss", "ev_tdlam_before", "entropy", "EpLenMean")})
            monitor.write_episodes(rets, lens)
        if log and rank == 0:
            log("iter %4d  steps %10d  eps %7d  EpLenMean %7.1f  (this iter %7.1f)  entropy %6.2f  meankl %.4f  surrgain %+.4f  ev %.3f  %.1fs"
                % (iters_so_far, timesteps_so_far, stats["EpThisIter"], stats["EpLenMean"], stats["EpLenMeanIter"], stats.get("entropy", float("nan")),
                   stats.get("meankl", float("nan")), stats.get("surrgain", float("nan")), stats["ev_tdlam_before"], stats["TimeElapsed"]))
    if progress is not None:
        progress.close(); monitor.close()
    return history


def runner(env, pi, timesteps_per_batch=1024, stochastic_policy=False, log=print):
    """`runner()` + `traj_1_generator()` of src/trpo.py:356-436 (`--task evaluate`), one trajectory per env of the batch at once: from
    `env.reset(); env.reset_model_init()` each env runs `pi.act(stochastic, ob)` -> `env.step(ac)` until its first `done` or until
    `timesteps_per_batch + 1` steps.  Returns (average length, average return) as the reference prints them, plus the per-trajectory
    arrays.  `pi` comes from `MlpPolicy.from_tf_checkpoint(path)` (= U.load_state) or `from_npz`."""
    n = env.num_envs
    dev = pi.device
    with torch.no_grad():
        ob = torch.zeros((n, 56), dtype=torch.float64, device=dev)
        env.reset("init", out=ob if ob.is_cuda else ob.numpy())
        alive = torch.ones(n, dtype=torch.bool, device=dev)
        ep_len = torch.zeros(n, dtype=torch.int64, device=dev); ep_ret = torch.zeros(n, dtype=torch.float64, device=dev)
        for t in range(int(timesteps_per_batch) + 1):
            ac, _ = pi.act(stochastic_policy, ob)
            res = env.step(ac if ac.is_cuda else ac.numpy())
            ob = torch.as_tensor(res[0], dtype=torch.float64, device=dev)
            rew = torch.as_tensor(res[1], dtype=torch.float64, device=dev); done = torch.as_tensor(res[2], device=dev).to(torch.bool)
            ep_ret += torch.where(alive, rew, torch.zeros_like(rew)); ep_len += alive.to(torch.int64)
            alive &= ~done
            if not bool(alive.any()):
                break
    lens, rets = ep_len.cpu().numpy(), ep_ret.cpu().numpy()
    log("stochastic policy:" if stochastic_policy else "deterministic policy:")
    log("Average length: %s" % (lens.sum() / len(lens)))
    log("Average return: %s" % (rets.sum() / len(rets)))
    return float(lens.mean()), float(rets.mean()), lens, rets
