"""Generates tests/golden/anchor/protocol_*.json: the reference's 2-worker TRPO protocol (src/trpo.py:338-353) replayed in the
oracle's physics with this repository's learner (tests/anchor.py).  ~15 min per run on one core.

    python tests/golden/gen/make_anchor.py <seed> [key=value ...]       e.g.  ... 0 pyramid_r_rescale=0
    python tests/golden/gen/make_anchor.py <seed> backend=gpu out=gpurun_out/anchor   (the same protocol on the HIP kernel, on an MI355X)
"""
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
sys.path.insert(0, ROOT)
from tests import anchor as AN  # noqa: E402


def main():
    seed = int(sys.argv[1])
    opts = {k: float(v) for k, v in (a.split("=") for a in sys.argv[2:] if not a.startswith(("iters=", "backend=", "out=")))}
    iters = [int(a.split("=")[1]) for a in sys.argv[2:] if a.startswith("iters=")]
    backend = ([a.split("=")[1] for a in sys.argv[2:] if a.startswith("backend=")] or ["oracle"])[0]       # backend=gpu: the HIP kernel (needs an MI355X)
    outdir = ([a.split("=")[1] for a in sys.argv[2:] if a.startswith("out=")] or [os.path.join(ROOT, "tests", "golden", "anchor")])[0]
    r = AN.run_reference_protocol(seed=seed, iterations=iters[0] if iters else 1900, log_every=100, backend=backend, **opts)
    tag = ("gpu_" if backend == "gpu" else "") + "seed%d" % seed + "".join("_%s%g" % (k, v) for k, v in sorted(opts.items()))
    os.makedirs(outdir, exist_ok=True)
    out = os.path.join(outdir, "protocol_%s.json" % tag)
    r["EpLenMean"] = [round(x, 2) for x in r["EpLenMean"]]
    json.dump(r, open(out, "w"))
    print("wrote", out)


if __name__ == "__main__":
    main()
