"""world_size-2 CPU (gloo) tests of the multi-GPU plumbing: env-range sharding and the per-horizon rollout all-gather."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from deepmimic_mujoco_amd.rollout import ROW, RolloutBlock, shard_range


def test_shard_range_partitions_exactly():
    for n, w in [(4096, 8), (32768, 8), (10, 3), (7, 8), (65536, 4)]:
        cuts = [shard_range(n, r, w) for r in range(w)]
        assert cuts[0][0] == 0 and cuts[-1][1] == n
        assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
        sizes = [b - a for a, b in cuts]
        assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, T, n):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_range(world * n, rank, world)
        blk = RolloutBlock(T, n)
        for t in range(T):
            env_ids = torch.arange(lo, hi, dtype=torch.float32)
            obs = env_ids[:, None] * 1000 + t + torch.arange(56, dtype=torch.float32)[None, :] / 100
            act = -obs[:, :28]
            full = blk.append(obs, act, env_ids + t, (env_ids % 2 == 0).float(), vpred=env_ids * 0.5)
            assert full == (t == T - 1)
        out = blk.gather()
        assert out.shape == (world, T, n, ROW)
        for r in range(world):
            rlo, _ = shard_range(world * n, r, world)
            ids = torch.arange(rlo, rlo + n, dtype=torch.float32)
            for t in (0, T - 1):
                assert torch.equal(out[r, t, :, 0], ids * 1000 + t)
                assert torch.equal(out[r, t, :, 56], -(ids * 1000 + t))
                assert torch.equal(out[r, t, :, 84], ids + t) and torch.equal(out[r, t, :, 85], (ids % 2 == 0).float())
                assert torch.equal(out[r, t, :, 86], ids * 0.5)
        assert blk.t == 0
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_rollout_allgather_world2_gloo():
    port = _free_port()
    mp.spawn(_worker, args=(2, port, 5, 6), nprocs=2, join=True)


def test_single_process_gather_is_identity():
    blk = RolloutBlock(2, 3)
    blk.append(torch.ones(3, 56), torch.zeros(3, 28), torch.ones(3), torch.zeros(3))
    out = blk.gather()
    assert out.shape == (1, 2, 3, ROW) and torch.equal(out[0, 0, :, :56], torch.ones(3, 56))


def _trpo_worker(rank, world, port, out_dir):
    """Two learner replicas on different data (the reference's `mpirun -np 2 python3 trpo.py`): all-mean'd gradients, Fisher
    products, value gradients and obs-filter moments must leave both replicas with identical parameters."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from deepmimic_mujoco_amd.policy import MlpPolicy
        from deepmimic_mujoco_amd.trpo import TrpoLearner
        from tests.test_trpo import _segment
        pi = MlpPolicy(seed=10 + rank)                      # different initial weights: rank 0's must win (trpo.py:182-186)
        L = TrpoLearner(pi, vf_batch_size=256)
        seg = _segment(pi, n=32, T=16, seed=100 + rank)     # different rollouts per rank
        seg["rew"] = -((seg["ac"] - 0.2) ** 2).mean(-1)
        st = L.update(seg)
        th = torch.cat([L.get_flat(), L.vfadam.getflat(), pi.ob_rms.sum.to(torch.float32), pi.ob_rms.count.reshape(1).to(torch.float32)])
        both = [torch.empty_like(th) for _ in range(world)]
        dist.all_gather(both, th)
        assert torch.equal(both[0], both[1]), float((both[0] - both[1]).abs().max())
        assert st["meankl"] <= 0.0151 and st["stepsize"] > 0
        if rank == 0:
            torch.save(th, os.path.join(out_dir, "theta.pt"))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_trpo_replicas_stay_in_sync_world2_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_trpo_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(str(tmp_path / "theta.pt"))


def _dbg_worker(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from deepmimic_mujoco_amd.rollout import DoubleBufferedGather
        T, n = 4, 3
        g = DoubleBufferedGather(T, n)
        assert len(g.blocks) == 2 and g.gathered[0].shape == (world * T, n, ROW)
        seen = []
        for t in range(5 * T):                                   # five horizons: every buffer is reused at least twice
            row = g.row(t)
            row[:] = float(1000 * rank + t)                      # rank- and step-dependent content
            k = g.commit(t)
            if k is not None:
                seen.append((t, k))
        g.drain()
        assert [k for _t, k in seen] == [0, 1, 0, 1, 0] and g.completed == 5
        # the last two horizons are still intact in the two gathered buffers: every rank's rows, in rank order
        for h, k in ((4, 0), (3, 1)):
            out = g.gathered[k].view(world, T, n, ROW)
            for r in range(world):
                for i in range(T):
                    assert torch.all(out[r, i] == float(1000 * r + h * T + i)), (h, k, r, i)
        # block-at-once producers (bench.py packs a whole horizon at its last step): same buffers, same gathers
        for h in range(5, 8):
            t = h * T + T - 1
            g.block(t)[:] = float(1000 * rank + 7 * h)
            assert g.commit(t) == h % 2
        g.drain()
        for h, k in ((7, 1), (6, 0)):
            out = g.gathered[k].view(world, T, n, ROW)
            for r in range(world):
                assert torch.all(out[r] == float(1000 * r + 7 * h))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_double_buffered_gather_world2_gloo():
    port = _free_port()
    mp.spawn(_dbg_worker, args=(2, port), nprocs=2, join=True)


def test_double_buffered_gather_single_process():
    from deepmimic_mujoco_amd.rollout import DoubleBufferedGather
    g = DoubleBufferedGather(4, 2, world=1)
    for t in range(9):
        g.row(t)[:] = t
        assert g.commit(t) is None
    g.drain()
    assert len(g.blocks) == 1 and float(g.blocks[0][0, 0, 0]) == 8.0


def test_bench_counts_distinct_gpus_without_pci_ids():
    """bench.py's start-up line and `config.distinct_gpus` (a SCALE record is read for them; src/train_mpi.sh:1 starts one worker per slot): PCI addresses when
    the torch build reports them, device uuids when it does not, and None — never a wrong count — when neither tells the ranks' devices apart."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(__file__), "..", "bench.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    assert m.distinct_gpus([[0, 5, 0, 0, 7], [0, 6, 0, 1, 8]]) == 2
    assert m.distinct_gpus([[0, 5, 0, 0, 7]] * 8) == 1                              # eight ranks sharing one device (the gloo test on the one-GPU box)
    assert m.distinct_gpus([[-1, -1, -1, 0, 7], [-1, -1, -1, 1, 8]]) == 2           # no PCI properties: uuids
    assert m.distinct_gpus([[-1, -1, -1, 0, 0], [-1, -1, -1, 1, 0]]) is None        # nothing to tell them apart: unknown, not "1"
