"""Every bundled clip on the device (run with -m gpu): a short oracle-parity rollout of the 5-term imitation reward at
`frame_skip="mocap"` for all 15 motions of src/mujoco/motions/humanoid3d_*.txt — looping clips wrap their frame cursor inside the
test (cycle counter + root shift), every `Loop: none` clip reaches its last frame and ends the episode there — plus two seeds of
the randomised HIP == oracle sweep (tools/fuzz_parity.py) so that the driver-run suite no longer leans on a tool run."""
import os
import subprocess
import sys

import numpy as np
import pytest

from deepmimic_mujoco_amd import _abi as A
from deepmimic_mujoco_amd.mocap import ALL_CLIPS
from tests import helpers as H
from tests.test_imitation import _imit_inputs, _rollout_vs_oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.mark.parametrize("clip", ALL_CLIPS)
def test_every_clip_imitation_rollout_matches_oracle(clip):
    from deepmimic_mujoco_amd import Batch
    sp, mc, T, P = _imit_inputs(clip)
    n, steps = 8, 8
    nsub = max(1, int(float(mc.dt) / 0.0166))                     # frame_skip="mocap"
    b = Batch(H.compiled_model(), mc.data_config, mc.data_vel, n, device=0, mocap_dt=float(mc.dt), imitation=(T, P))
    worst, cyc = _rollout_vs_oracle(b, n, steps=steps, nsub=nsub, seed=7, clip=clip)     # envs 0 / 1 start at frames F-3 / F-2
    assert worst < 1e-9, "%s: %.3e" % (clip, worst)
    fi = b.get(A.F_FRAME_IDX)
    if mc.loop == "none":
        assert P[15] == 0.0 and np.all(cyc == 0) and fi[0] == len(T) - 1 and fi[1] == len(T) - 1      # held on the last frame (done was compared step by step)
    else:
        assert P[15] == 1.0 and cyc[0] == 1 and cyc[1] == 1       # wrapped once
    b.close()


@pytest.mark.parametrize("clip", ALL_CLIPS)
def test_every_clip_through_one_horizon_launch_matches_oracle(clip):
    """The same rollouts — 5-term imitation reward, `frame_skip="mocap"`, wrapping and `Loop: none` clips — as ONE dm_batch_rollout launch on the
    packed path (four environments per wavefront, every wave through all steps at its own pace; 10 envs = two full waves + two spare
    slots): every row of obs / reward / done against the oracle stepped env by env, frame cursors and cycle counters at the end."""
    import torch
    from deepmimic_mujoco_amd import Batch
    from oracle import oracle as O
    sp, mc, T, P = _imit_inputs(clip)
    n, steps = 10, 8
    nsub = max(1, int(float(mc.dt) / 0.0166))
    F = len(T)
    b = Batch(H.compiled_model(), mc.data_config, mc.data_vel, n, device=0, mocap_dt=float(mc.dt), imitation=(T, P))
    b.set_option(A.OPT_REWARD_MODE, 3); b.set_option(A.OPT_PACKED, 1); b.set_option(106, 1)
    rng = np.random.RandomState(7)
    idx = np.concatenate([[F - 3, F - 2], rng.randint(0, F, size=n - 2)]).astype(np.int32)
    q = mc.data_config[idx].copy(); v = mc.data_vel[idx].copy()
    q[n // 2:, 7:] += 0.05 * rng.randn(n - n // 2, 28)
    b.set(A.F_QACC_WARMSTART, np.zeros((n, 34))); b.set(A.F_TIME, np.zeros(n))
    b.set_state(q, v, frame_idx=idx)
    acts = rng.randn(steps + 1, n, 28) * 0.3
    dev = "cuda:0"
    ac = torch.as_tensor(acts, dtype=torch.float64, device=dev).contiguous()
    ob = torch.zeros((steps, n, 56), dtype=torch.float64, device=dev); rew = torch.zeros((steps, n), dtype=torch.float64, device=dev)
    dn = torch.zeros((steps, n), dtype=torch.uint8, device=dev)
    b.rollout(ac, (ob, rew, dn), nsub)
    b.join(); b.sync()
    ob, rew, dn = ob.cpu().numpy(), rew.cpu().numpy(), dn.cpu().numpy()
    om = H.oracle_model()
    ods = [O.Data(om) for _ in range(n)]
    for e in range(n):
        ods[e].reset(); ods[e].set_state(q[e], v[e])
    fidx = idx.astype(int).copy(); cyc = np.zeros(n, int)
    worst = 0.0
    for t in range(steps):
        for e in range(n):
            o, r, d, fidx[e], cyc[e] = O.env_step_imitation(om, ods[e], acts[t, e], nsub, T, P, fidx[e], cyc[e])
            worst = max(worst, abs(rew[t, e] - r), H.rel_err(ob[t, e], o))
            assert bool(dn[t, e]) == d, (clip, t, e)
    assert worst < 1e-9, "%s: %.3e" % (clip, worst)
    assert np.array_equal(b.get(A.F_FRAME_IDX), fidx.astype(np.int32)) and np.array_equal(b.get(A.F_CYCLE), cyc.astype(np.int32))
    if mc.loop == "none":
        assert fidx[0] == F - 1 and fidx[1] == F - 1
    else:
        assert cyc[0] == 1 and cyc[1] == 1
    b.close()


def test_at_least_one_clip_of_each_loop_kind_is_bundled():
    kinds = {H.mocap(c).loop for c in ALL_CLIPS}
    assert kinds == {"wrap", "none"}, kinds


@pytest.mark.parametrize("seed", [101, 102])
def test_fuzz_parity_seed(seed):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "48", str(seed)], capture_output=True, text=True,
                         timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "fuzz seed %d: ok" % seed in out.stdout
