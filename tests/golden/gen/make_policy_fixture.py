#!/usr/bin/env python3
"""Copies the reference's one shipped policy checkpoint (DATA: a tf.train.Saver tensor bundle, src/trpo.py:268-270) and
extracts the episode-length curve of the run that produced it (DATA: src/log_tmp/.../log.txt) into fixtures.

    python tests/golden/gen/make_policy_fixture.py            (needs /root/reference; run in the build container only)

Outputs
  tests/golden/ckpt/trpo-walk-0.index, .data-00000-of-00001   the bundle, byte for byte (weights of 'pi' and 'oldpi')
  tests/golden/trpo_walk0_log.npz                             EpLenMean / EpRewMean / TimestepsSoFar / entropy per iteration
  tests/golden/policy_forward_golden.npz                      float64 numpy restatement of mlp_policy_trpo.py:35-46 on seeded inputs
"""
import os
import re
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", "..", ".."))
REF = os.environ.get("DM_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)


def main():
    src = os.path.join(REF, "src/checkpoint_tmp/DeepMimic/trpo-walk-0/DeepMimic")
    dst = os.path.join(ROOT, "tests/golden/ckpt")
    os.makedirs(dst, exist_ok=True)
    for f in ("trpo-walk-0.index", "trpo-walk-0.data-00000-of-00001"):
        shutil.copyfile(os.path.join(src, f), os.path.join(dst, f))
        os.chmod(os.path.join(dst, f), 0o644)

    cols = {"EpLenMean": [], "EpRewMean": [], "TimestepsSoFar": [], "entropy": [], "EpisodesSoFar": []}
    pat = re.compile(r"^\|\s*(\w+)\s*\|\s*([-+0-9.eE]+)\s*\|")
    for line in open(os.path.join(REF, "src/log_tmp/DeepMimic/trpo-walk-0/log.txt")):
        m = pat.match(line)
        if m and m.group(1) in cols:
            cols[m.group(1)].append(float(m.group(2)))
    np.savez(os.path.join(ROOT, "tests/golden/trpo_walk0_log.npz"), **{k: np.asarray(v) for k, v in cols.items()})

    # forward pass in float64 numpy straight from the checkpoint bytes (independent of policy.py / torch)
    from deepmimic_mujoco_amd.tf_checkpoint import load_checkpoint
    d = load_checkpoint(os.path.join(dst, "trpo-walk-0"), scope="pi")
    rng = np.random.RandomState(7)
    mean = (d["obfilter/runningsum"] / d["obfilter/count"]).astype(np.float32)
    std = np.sqrt(np.maximum((d["obfilter/runningsumsq"] / d["obfilter/count"]).astype(np.float32) - mean ** 2, 1e-2))
    ob = (mean + std * rng.randn(64, 56) * 2.5).astype(np.float64)       # some coordinates beyond the +-5 clip
    z = np.clip((ob - mean) / std, -5, 5)
    lin = lambda x, n: x @ d[n + "/w"].astype(np.float64) + d[n + "/b"].astype(np.float64)
    v = lin(np.tanh(lin(np.tanh(lin(z, "vffc1")), "vffc2")), "vffinal")[:, 0]
    a = lin(np.tanh(lin(np.tanh(lin(z, "polfc1")), "polfc2")), "polfinal")
    np.savez(os.path.join(ROOT, "tests/golden/policy_forward_golden.npz"), ob=ob, mean=a, vpred=v)
    print("EpLenMean last", cols["EpLenMean"][-1], "iterations", len(cols["EpLenMean"]))


if __name__ == "__main__":
    main()
