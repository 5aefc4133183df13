#!/usr/bin/env python3
"""Fuzz the HIP path against the CPU oracle (TEST TOOL; needs an MI355X): seeded states around every bundled clip — mocap frames,
perturbed poses, limits violated, feet / hands / torso in the floor, tumbling — one forward evaluation each, every stage compared
(tests/helpers.compare_forward: M, bias, J, R, aref, b, forces, qacc, contact geom lists, row counts, PGS sweep counts), then short
rollouts.  usage: python tools/fuzz_parity.py [states-per-clip] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from deepmimic_mujoco_amd import Batch, _abi as A  # noqa: E402
from deepmimic_mujoco_amd.mocap import ALL_CLIPS  # noqa: E402
from tests import helpers as H  # noqa: E402


def main():
    per = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    om = H.oracle_model()
    worst_all = {}
    t0 = time.time()
    nstates = nsteps = 0
    for ci, clip in enumerate(ALL_CLIPS):
        mc = H.mocap(clip)
        idx, q, v, ws, ctrl = H.varied_states(per, seed=seed * 1000 + ci, clip=clip)
        rng = np.random.RandomState(seed * 77 + ci)
        for e in range(0, per, 5):                       # a share of deep-penetration / lying poses (many rows, overflow strip)
            q[e, 2] = 0.05 + 0.3 * rng.rand(); v[e] *= 0.3
        b = Batch(H.compiled_model(), mc.data_config, mc.data_vel, per, device=0, mocap_dt=float(mc.dt))
        worst = H.compare_forward(b, om, idx, q, v, ws, ctrl)
        w2, nd = H.compare_rollout(b, om, idx, q, v, steps=4, seed=seed + ci, clip=clip)
        nefc = b.get(A.F_NEFC)
        b.close()
        nstates += per; nsteps += 4 * per
        for k, val in worst.items():
            worst_all[k] = max(worst_all.get(k, 0.0), val)
        worst_all["rollout"] = max(worst_all.get("rollout", 0.0), w2)
        print("%-16s forward ok (worst %.1e), rollout ok (worst %.1e, %d done), nefc after rollout: mean %.1f max %d"
              % (clip, max(worst.values()), w2, nd, nefc.mean(), nefc.max()), flush=True)
    print("fuzz: %d states, %d env-steps across %d clips in %.0f s; worst relative errors: %s"
          % (nstates, nsteps, len(ALL_CLIPS), time.time() - t0, {k: "%.1e" % v for k, v in sorted(worst_all.items())}))


if __name__ == "__main__":
    main()
