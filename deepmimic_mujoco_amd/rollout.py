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
