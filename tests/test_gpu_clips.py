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


def test_at_least_one_clip_of_each_loop_kind_is_bundled():
    kinds = {H.mocap(c).loop for c in ALL_CLIPS}
    assert kinds == {"wrap", "none"}, kinds


@pytest.mark.parametrize("seed", [101, 102])
def test_fuzz_parity_seed(seed):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "48", str(seed)], capture_output=True, text=True,
                         timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "fuzz seed %d: ok" % seed in out.stdout
