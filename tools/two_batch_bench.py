"""Round 6, review item 8b: BASELINE configs[2] stepped CLOSED LOOP (one launch set per call, nothing queued) as TWO interleaved 4 096-env batches on the
four-per-wave kernel — 2 x 1 024 packed waves on 1 024 SIMDs, the regime in which one 8 192-env batch steps at ~20 M env-steps/s — against one 4 096-env batch on
the kernel DPVecEnv picks for it (k_step_narrow) and on k_step_packed.  What a host-side policy that drives two env sets gets (the VecEnv contract:
src/utils/vec_env/__init__.py:26-100, dummy_vec_env.py:45-56): while set A's slowest waves drain, set B's fill the SIMDs they left.
    python tools/two_batch_bench.py [steps] [repeats]
Prints one line per form: aggregate env-steps/s and the rate at which EACH 4 096-env set advances."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from deepmimic_mujoco_amd import DPVecEnv  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n = 4096
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(3)


def make(packed, seed, streams=None):
    env = DPVecEnv(n, motion="walk", device=0, reward="imitation", autoreset="rsi", seed=seed, packed=packed, frame_skip=1)
    env.reset("rsi")
    acts = torch.randn((64, n, 28), generator=g, dtype=torch.float64, device=dev) * 0.9
    out = [(torch.empty((n, 56), dtype=torch.float64, device=dev), torch.empty(n, dtype=torch.float64, device=dev), torch.empty(n, dtype=torch.uint8, device=dev)) for _ in range(2)]
    return env, acts, out


def run(envs, label, own_streams=False):
    sts = [torch.cuda.Stream(device=dev) for _ in envs] if own_streams else None

    def window():
        for t in range(steps):
            for i, (env, acts, out) in enumerate(envs):
                if sts:
                    with torch.cuda.stream(sts[i]):
                        env.step(acts[t % 64], out=out[t & 1])
                else:
                    env.step(acts[t % 64], out=out[t & 1])
        for env, _, _ in envs:
            env.batch.join()
        torch.cuda.synchronize()
    window()                                            # untimed
    best = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); window(); best.append(time.perf_counter() - t0)
    el = sorted(best)[len(best) // 2]
    agg = len(envs) * n * steps / el
    print("%-64s aggregate %6.2f M env-steps/s   each 4096-env set advances at %6.2f M/s   (%.3f ms per round of steps)" % (label, agg / 1e6, n * steps / el / 1e6, 1e3 * el / steps), flush=True)
    return agg


a = make(False, 1)
run([a], "one 4096-env batch, k_step_narrow (what DPVecEnv picks)")
a[0].close()
b = make(True, 1)
run([b], "one 4096-env batch, k_step_packed")
c = make(True, 2)
run([b, c], "TWO 4096-env batches interleaved, k_step_packed, one stream")
run([b, c], "TWO 4096-env batches interleaved, k_step_packed, a stream each", own_streams=True)
b[0].close(); c[0].close()
d = make(False, 1); e = make(False, 2)
run([d, e], "TWO 4096-env batches interleaved, k_step_narrow, a stream each", own_streams=True)
d[0].close(); e[0].close()
f = DPVecEnv(2 * n, motion="walk", device=0, reward="imitation", autoreset="rsi", seed=1, packed=True, frame_skip=1); f.reset("rsi")
acts = torch.randn((64, 2 * n, 28), generator=g, dtype=torch.float64, device=dev) * 0.9
out = [(torch.empty((2 * n, 56), dtype=torch.float64, device=dev), torch.empty(2 * n, dtype=torch.float64, device=dev), torch.empty(2 * n, dtype=torch.uint8, device=dev)) for _ in range(2)]
n2 = n
n = 2 * n2
run([(f, acts, out)], "one 8192-env batch, k_step_packed (reference point)")
