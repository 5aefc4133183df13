"""Batched MLP policy + value function of the reference's TRPO learner, as plain torch tensors on the env's device.

Mirror of `src/mlp_policy_trpo.py:13-79` (class MlpPolicy) and `src/utils/misc_util.py:32-70` (RunningMeanStd):

    obz   = clip((ob - ob_rms.mean) / ob_rms.std, -5, 5)                    (mlp_policy_trpo.py:35)
    vpred = vffinal(tanh(vffc2(tanh(vffc1(obz)))))[:, 0]                    (:37-39)
    mean  = polfinal(tanh(polfc2(tanh(polfc1(obz)))))                       (:41-46)
    ac    = mean + exp(logstd) * N(0, 1)   if stochastic else mean          (:57-58, distributions.py DiagGaussianPd)

`act(stochastic, ob)` takes the whole [N, 56] observation batch that `dm_batch_step` wrote on the device and returns
([N, 28] float64 actions ready for the next `dm_batch_step`, [N] vpred) without leaving the device or the stream.
On a GPU the whole step is ONE launch of the hand-written `k_policy_act` kernel of libdmenv.so (`dm_policy_act`,
csrc/policy_kernel.h: normalisation, both MLPs, Gaussian sample; float32 like the reference's TF graph) instead of ~15
launch-bound library calls; `forward()` — what the learner differentiates — and the CPU path are plain torch ops, and the
two agree to float32 rounding (tests/test_gpu_rollout.py).

Weights are interchangeable with the reference's checkpoints: `MlpPolicy.from_tf_checkpoint(prefix)` reads a
`tf.train.Saver` bundle (scope 'pi'), `state_dict()/load_state_dict()` use the reference's variable names.
"""
import math

import numpy as np
import torch

from .tf_checkpoint import load_checkpoint


class RunningMeanStd:
    """src/utils/misc_util.py:32-70.  Sums in float64; count and sumsq start at epsilon = 1e-2; the variance is floored
    at 1e-2 before the sqrt.  `update` all-reduces (sum, sumsq, count) over the process group like the reference's
    MPI.Allreduce (:69) when torch.distributed is initialised."""

    def __init__(self, shape, device="cpu", epsilon=1e-2):
        self.shape = tuple(shape) if not isinstance(shape, int) else (shape,)
        self.sum = torch.zeros(self.shape, dtype=torch.float64, device=device)
        self.sumsq = torch.full(self.shape, epsilon, dtype=torch.float64, device=device)
        self.count = torch.full((), epsilon, dtype=torch.float64, device=device)
        self._refresh()

    def _refresh(self):
        mean = (self.sum / self.count).to(torch.float32)
        var = (self.sumsq / self.count).to(torch.float32) - mean * mean
        std = torch.sqrt(torch.clamp(var, min=1e-2))
        if getattr(self, "mean", None) is not None and self.mean.shape == mean.shape and self.mean.device == mean.device:
            self.mean.copy_(mean); self.std.copy_(std)       # in place: a captured graph of the policy keeps reading these buffers
        else:
            self.mean, self.std = mean, std

    def update(self, x, group=None):
        x = x.reshape(-1, *self.shape).to(torch.float64)
        n = self.sum.numel()
        add = torch.cat([x.sum(0).reshape(-1), (x * x).sum(0).reshape(-1),
                         torch.tensor([float(x.shape[0])], dtype=torch.float64, device=x.device)])
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(add, op=dist.ReduceOp.SUM, group=group)
        self.sum += add[:n].reshape(self.shape)
        self.sumsq += add[n:2 * n].reshape(self.shape)
        self.count += add[2 * n]
        self._refresh()


def _normc(gen, fan_in, fan_out, std, device):
    """U.normc_initializer (src/utils/tf_util.py:98-103): Gaussian columns rescaled to L2 norm `std`."""
    w = torch.randn((fan_in, fan_out), generator=gen, dtype=torch.float32)
    w *= std / torch.sqrt((w * w).sum(0, keepdim=True))
    return w.to(device)


_LAYERS = ("vffc1", "vffc2", "vffinal", "polfc1", "polfc2", "polfinal")


class MlpPolicy:
    """`MlpPolicy(name, ob_space, ac_space, hid_size=100, num_hid_layers=2)` of the reference, for N envs at once."""

    recurrent = False

    def __init__(self, ob_dim=56, ac_dim=28, hid_size=100, num_hid_layers=2, device="cpu", seed=0):
        if num_hid_layers != 2:
            raise ValueError("the reference trains 2 hidden layers (src/trpo.py:436); got %d" % num_hid_layers)
        self.ob_dim, self.ac_dim, self.hid_size = ob_dim, ac_dim, hid_size
        self.device = torch.device(device)
        gen = torch.Generator(); gen.manual_seed(seed)
        self.ob_rms = RunningMeanStd((ob_dim,), device=self.device)
        p = {}
        dims = {"vffc1": (ob_dim, hid_size, 1.0), "vffc2": (hid_size, hid_size, 1.0), "vffinal": (hid_size, 1, 1.0),
                "polfc1": (ob_dim, hid_size, 1.0), "polfc2": (hid_size, hid_size, 1.0), "polfinal": (hid_size, ac_dim, 0.01)}
        for name in _LAYERS:
            i, o, s = dims[name]
            p[name + "/w"] = _normc(gen, i, o, s, self.device)
            p[name + "/b"] = torch.zeros(o, dtype=torch.float32, device=self.device)
        p["logstd"] = torch.zeros((1, ac_dim), dtype=torch.float32, device=self.device)
        self.params = p
        self._noise_gen = None
        self._packed = None        # float32 device block for dm_policy_act; rebuilt when `_dirty`
        self._dirty = True
        self._seed, self._counter = 0, 0
        self.native = True         # use the HIP kernel for act() on CUDA tensors (False: torch ops everywhere)

    # ---- weights ----------------------------------------------------------------------------------------------------
    def state_dict(self):
        d = {k: v.detach().cpu().numpy().copy() for k, v in self.params.items()}
        d["obfilter/runningsum"] = self.ob_rms.sum.cpu().numpy()
        d["obfilter/runningsumsq"] = self.ob_rms.sumsq.cpu().numpy()
        d["obfilter/count"] = self.ob_rms.count.cpu().numpy()
        return d

    def load_state_dict(self, d):
        for k in self.params:
            v = np.asarray(d[k])
            if tuple(v.shape) != tuple(self.params[k].shape):
                raise ValueError("%s: shape %s != %s" % (k, v.shape, tuple(self.params[k].shape)))
            rg = self.params[k].requires_grad
            self.params[k] = torch.as_tensor(v, dtype=torch.float32).to(self.device).contiguous().clone().requires_grad_(rg)
        self.ob_rms.sum = torch.as_tensor(np.asarray(d["obfilter/runningsum"]), dtype=torch.float64).to(self.device)
        self.ob_rms.sumsq = torch.as_tensor(np.asarray(d["obfilter/runningsumsq"]), dtype=torch.float64).to(self.device)
        self.ob_rms.count = torch.as_tensor(np.asarray(d["obfilter/count"]), dtype=torch.float64).to(self.device)
        self.ob_rms._refresh()
        self._dirty = True
        return self

    @classmethod
    def from_tf_checkpoint(cls, prefix, scope="pi", device="cpu"):
        """`U.load_state(load_model_path)` (src/trpo.py:365): restore the 'pi' scope of a tf.train.Saver bundle."""
        d = load_checkpoint(prefix, scope=scope)
        pol = cls(ob_dim=d["polfc1/w"].shape[0], ac_dim=d["polfinal/w"].shape[1], hid_size=d["polfc1/w"].shape[1], device=device)
        return pol.load_state_dict(d)

    @classmethod
    def from_npz(cls, path, device="cpu"):
        d = dict(np.load(path))
        pol = cls(ob_dim=d["polfc1/w"].shape[0], ac_dim=d["polfinal/w"].shape[1], hid_size=d["polfc1/w"].shape[1], device=device)
        return pol.load_state_dict(d)

    def save_npz(self, path):
        np.savez(path, **self.state_dict())

    def save_tf_checkpoint(self, prefix, old=None):
        """Write the policy as the `tf.train.Saver` bundle the reference saves (src/trpo.py:220-224) and restores
        (`U.load_state`, src/utils/tf_util.py:314-319; `--task evaluate`, src/trpo.py:367): variables `pi/...` and `oldpi/...`
        (the learner's two MlpPolicy scopes, :128-129) with the reference's names, dtypes and shapes — float32 weights, `logstd`
        [1, ac], float64 filter sums, scalar count.  `old`: state dict for the `oldpi` scope (default: a copy of `pi`, which is
        what `assign_old_eq_new` leaves behind at every update, :247)."""
        from .tf_checkpoint import save_checkpoint
        def cast(d):
            out = {}
            for k, v in d.items():
                v = np.asarray(v)
                out[k] = v.astype(np.float64) if k.startswith("obfilter/") else v.astype(np.float32)
            out["obfilter/count"] = np.asarray(out["obfilter/count"], dtype=np.float64).reshape(())
            return out
        cur = cast(self.state_dict()); prev = cast(old) if old is not None else cur
        tensors = {"pi/" + k: v for k, v in cur.items()}
        tensors.update({"oldpi/" + k: v for k, v in prev.items()})
        return save_checkpoint(prefix, tensors)

    # ---- forward ----------------------------------------------------------------------------------------------------
    def _obz(self, ob):
        ob = ob.to(torch.float32)
        return torch.clamp((ob - self.ob_rms.mean) / self.ob_rms.std, -5.0, 5.0)

    def forward_value(self, ob, z=None):
        """ob [N, ob_dim] -> vpred [N] f32 (the value net alone: the learner's value fit needs nothing else)."""
        p = self.params
        if z is None:
            z = self._obz(ob)
        h = torch.tanh(torch.addmm(p["vffc1/b"], z, p["vffc1/w"]))
        h = torch.tanh(torch.addmm(p["vffc2/b"], h, p["vffc2/w"]))
        return torch.addmm(p["vffinal/b"], h, p["vffinal/w"])[:, 0]

    def forward_mean(self, ob, z=None):
        """ob [N, ob_dim] -> action mean [N, ac_dim] f32 (the policy net alone: surrogate, KL and Fisher products)."""
        p = self.params
        if z is None:
            z = self._obz(ob)
        h = torch.tanh(torch.addmm(p["polfc1/b"], z, p["polfc1/w"]))
        h = torch.tanh(torch.addmm(p["polfc2/b"], h, p["polfc2/w"]))
        return torch.addmm(p["polfinal/b"], h, p["polfinal/w"])

    def forward(self, ob):
        """ob [N, ob_dim] -> (mean [N, ac_dim] f32, vpred [N] f32)."""
        z = self._obz(ob)
        return self.forward_mean(ob, z), self.forward_value(ob, z)

    def seed(self, seed):
        self._noise_gen = torch.Generator(device=self.device)
        self._noise_gen.manual_seed(int(seed))
        self._seed, self._counter = int(seed), 0

    # ---- native path -------------------------------------------------------------------------------------------------------
    def mark_dirty(self):
        """Parameters or obs-filter moments changed in place (the learner calls this): repack before the next native act()."""
        self._dirty = True

    def pack(self):
        """Flat float32 block in the layout of csrc/policy_kernel.h (dm_policy_weight_count() floats)."""
        p = self.params
        parts = [self.ob_rms.mean, self.ob_rms.std,
                 p["polfc1/w"], p["polfc1/b"], p["polfc2/w"], p["polfc2/b"], p["polfinal/w"], p["polfinal/b"], p["logstd"],
                 p["vffc1/w"], p["vffc1/b"], p["vffc2/w"], p["vffc2/b"], p["vffinal/w"], p["vffinal/b"]]
        with torch.no_grad():
            flat = torch.cat([t.detach().reshape(-1).to(torch.float32) for t in parts])
        if self._packed is None or self._packed.shape != flat.shape:
            self._packed = flat.contiguous()
        else:
            self._packed.copy_(flat)
        self._dirty = False
        return self._packed

    def _act_native(self, stochastic, ob, out, vpred_out):
        import ctypes as C
        from . import _abi as A
        L = A.load()
        n = ob.shape[0]
        if self._dirty or self._packed is None:
            self.pack()
            if self._packed.numel() != L.dm_policy_weight_count():
                raise RuntimeError("packed policy has %d floats, the kernel expects %d" % (self._packed.numel(), L.dm_policy_weight_count()))
        if out is None:
            out = torch.empty((n, self.ac_dim), dtype=torch.float64, device=ob.device)
        if vpred_out is None:
            vpred_out = torch.empty(n, dtype=torch.float32, device=ob.device)
        self._counter += 1
        stream = torch.cuda.current_stream(ob.device).cuda_stream
        A.check(L.dm_policy_act(C.c_void_p(self._packed.data_ptr()), C.c_void_p(ob.data_ptr()), C.c_void_p(out.data_ptr()),
                                C.c_void_p(vpred_out.data_ptr()), n, 1 if stochastic else 0, self._seed & (2 ** 64 - 1), self._counter,
                                C.c_void_p(stream)), L)
        return out, vpred_out

    def act(self, stochastic, ob, out=None, vpred_out=None):
        """mlp_policy_trpo.py:63-65 for a batch: returns (ac [N, ac_dim] float64, vpred [N] float32).
        `out` (float64 [N, ac_dim]) receives the action in place when given (the buffer handed to dm_batch_step);
        `vpred_out` (float32 [N]) likewise receives the value prediction."""
        single = ob.dim() == 1
        if single:
            ob = ob[None]
        if (self.native and ob.is_cuda and ob.dtype == torch.float64 and ob.is_contiguous() and self.ob_dim == 56 and self.ac_dim == 28
                and self.hid_size == 100 and (out is None or (out.is_contiguous() and out.dtype == torch.float64))
                and (vpred_out is None or (vpred_out.is_contiguous() and vpred_out.dtype == torch.float32))):
            ac, vpred = self._act_native(stochastic, ob, out, vpred_out)
            return (ac[0], vpred[0]) if single else (ac, vpred)
        mean, vpred = self.forward(ob)
        if stochastic:
            noise = torch.randn(mean.shape, dtype=torch.float32, device=mean.device, generator=self._noise_gen)
            ac = torch.addcmul(mean, self._std(), noise)
        else:
            ac = mean
        if vpred_out is not None:
            vpred_out.copy_(vpred)
            vpred = vpred_out
        if out is not None:
            out.copy_(ac)
            ac = out
        else:
            ac = ac.to(torch.float64)
        return (ac[0], vpred[0]) if single else (ac, vpred)

    def _std(self):
        return torch.exp(self.params["logstd"])

    # ---- distribution (src/distributions.py:220-243 DiagGaussianPd) ---------------------------------------------------
    def neglogp(self, ob, ac):
        mean, _ = self.forward(ob)
        logstd = self.params["logstd"]
        z = (ac.to(torch.float32) - mean) / torch.exp(logstd)
        return 0.5 * (z * z).sum(-1) + 0.5 * math.log(2.0 * math.pi) * self.ac_dim + logstd.sum(-1)

    def entropy(self):
        return float((self.params["logstd"].detach() + 0.5 * math.log(2.0 * math.pi * math.e)).sum())
