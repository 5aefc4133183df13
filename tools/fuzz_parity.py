#!/usr/bin/env python3
"""Fuzz the HIP path against the CPU oracle (TEST TOOL; needs an MI355X): seeded states around every bundled clip — mocap frames,
perturbed poses, limits violated, feet / hands / torso in the floor, tumbling — one forward evaluation each, every stage compared
(tests/helpers.compare_forward: M, bias, J, R, aref, b, forces, qacc, contact geom lists, row counts, PGS sweep counts), then short
rollouts.  usage: [DM_FUZZ_KEEP=dir] python tools/fuzz_parity.py [states-per-clip] [seed]  (failing inputs are kept in dir)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from deepmimic_mujoco_amd import Batch, _abi as A  # noqa: E402
from deepmimic_mujoco_amd.mocap import ALL_CLIPS  # noqa: E402
from tests import helpers as H  # noqa: E402


FAILURES = []
KEEP_DIR = os.environ.get("DM_FUZZ_KEEP", "")


def _keep(name, arr):
    if KEEP_DIR:
        os.makedirs(KEEP_DIR, exist_ok=True)
        np.save(os.path.join(KEEP_DIR, name + ".npy"), arr)


def forward(batch, om, idx, q, v, ws, ctrl, tag):
    """compare_forward over the whole batch; every failing state is named and kept."""
    bad = []
    worst = H.compare_forward(batch, om, idx, q, v, ws, ctrl, failures=bad)
    for e, msg in bad:
        FAILURES.append("forward %s env %d: %s" % (tag, e, msg.splitlines()[0][:200]))
        _keep("forward_%s_e%d" % (tag, e), np.concatenate([q[e], v[e], ws[e], ctrl[e]]))
    return worst


def rollout(batch, om, q, v, steps, seed, tol=1e-9, tag=""):
    """Lock-step rollout, every env against its own oracle instance.  The step map is discontinuous (a contact entering its margin, the
    row capacity cutting the contact list, PGS stopping one sweep earlier): in extreme states (a body sunk into the floor, 60 rows)
    rounding-level differences land on opposite sides — the ORACLE ITSELF answers 0.8 apart for inputs 1e-15 apart there.  So a
    mismatch is only an error if the device's result matches none of the oracle's branches in a 1e-15 neighbourhood of the device's
    own state of that step; otherwise the oracle is re-synchronised on the matching branch and the event is counted."""
    from oracle import oracle as O
    n = q.shape[0]
    batch.set(A.F_QACC_WARMSTART, np.zeros((n, 34))); batch.set(A.F_TIME, np.zeros(n))
    batch.set_state(q, v)
    ods = [O.Data(om) for _ in range(n)]
    for e in range(n):
        ods[e].reset(); ods[e].set_state(q[e], v[e])
    rng = np.random.RandomState(seed)
    worst = 0.0; ndone = nsens = 0
    for t in range(steps):
        a = rng.randn(n, 28) * 0.9
        q0 = batch.get(A.F_QPOS); v0 = batch.get(A.F_QVEL); w0 = batch.get(A.F_QACC_WARMSTART)
        obs, rew, done = batch.step(a, 1)[:3]
        for e in range(n):
            o, r, d, _ = ods[e].env_step(a[e])
            err = H.rel_err(obs[e], o)
            if err >= tol or bool(done[e]) != d:
                # restart the oracle from the device's own state of this step; if that still disagrees, probe a 1e-15 neighbourhood of
                # it: at a discontinuity the oracle's own answers there split into branches, one of which is the device's
                prs = np.random.RandomState(1000 * t + e)
                ok = False
                for k in range(24):
                    dq = q0[e].copy()
                    if k:
                        dq[7:] += 1e-15 * prs.randn(28)
                    od = O.Data(om); od.reset(); od.set("qacc_warmstart", w0[e]); od.set_state(dq, v0[e])
                    o, r, d, _ = od.env_step(a[e])
                    err = H.rel_err(obs[e], o)
                    if err < tol and bool(done[e]) == d:
                        ok = True
                        break
                if not ok:                              # keep the inputs for the wave testbench, go on from the device's result
                    FAILURES.append("rollout env %d step %d: HIP step matches no branch of the oracle around the same inputs (%.3e)" % (e, t, err))
                    _keep("rollout_%s_e%d_t%d" % (tag, e, t), np.concatenate([q0[e], v0[e], w0[e], a[e], obs[e]]))
                    od = O.Data(om); od.reset(); od.set("qacc_warmstart", batch.get(A.F_QACC_WARMSTART)[e])
                    od.set_state(batch.get(A.F_QPOS)[e], batch.get(A.F_QVEL)[e])
                ods[e] = od; nsens += 1
            worst = max(worst, err); ndone += int(d)
    return worst, ndone, nsens


def main():
    per = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    om = H.oracle_model()
    worst_all = {}
    sens_total = 0
    t0 = time.time()
    nstates = nsteps = 0
    for ci, clip in enumerate(ALL_CLIPS):
        mc = H.mocap(clip)
        idx, q, v, ws, ctrl = H.varied_states(per, seed=seed * 1000 + ci, clip=clip)
        rng = np.random.RandomState(seed * 77 + ci)
        for e in range(0, per, 5):                       # a share of deep-penetration / lying poses (many rows, overflow strip)
            q[e, 2] = 0.05 + 0.3 * rng.rand(); v[e] *= 0.3
        b = Batch(H.compiled_model(), mc.data_config, mc.data_vel, per, device=0, mocap_dt=float(mc.dt))
        tag = "%s_s%d" % (clip, seed)
        worst = forward(b, om, idx, q, v, ws, ctrl, tag)
        w2, nd, nsens = rollout(b, om, q, v, steps=4, seed=seed + ci, tag=tag)
        sens_total += nsens
        nefc = b.get(A.F_NEFC)
        b.close()
        nstates += per; nsteps += 4 * per
        for k, val in worst.items():
            worst_all[k] = max(worst_all.get(k, 0.0), val)
        worst_all["rollout"] = max(worst_all.get("rollout", 0.0), w2)
        print("%-16s forward ok (worst %.1e), rollout ok (worst %.1e, %d done), nefc after rollout: mean %.1f max %d"
              % (clip, max(worst.values()), w2, nd, nefc.mean(), nefc.max()), flush=True)
    print("fuzz: %d states, %d env-steps across %d clips in %.0f s; worst relative errors: %s; env-steps re-synchronised at a "
          "discontinuity of the step map: %d"
          % (nstates, nsteps, len(ALL_CLIPS), time.time() - t0, {k: "%.1e" % v for k, v in sorted(worst_all.items())}, sens_total))
    for f in FAILURES:
        print("FAIL", f)
    print("fuzz seed %d: %s" % (seed, "%d FAILURES" % len(FAILURES) if FAILURES else "ok"))
    sys.exit(1 if FAILURES else 0)


if __name__ == "__main__":
    main()
