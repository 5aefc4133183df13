"""Build libdmenv.so (HIP, gfx950) in-tree with hipcc.  Usage: python -m deepmimic_mujoco_amd.csrc.build [--force]"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
SOURCES = ["dmenv.hip", "env_kernel.h", "env_step.h", "model_host.h", "topology.h", "wave.h", "policy_kernel.h", "vf_kernel.h", "pg_kernel.h", "slot_kernel.h", "slot_step.h"]
OUT = os.path.join(HERE, "libdmenv.so")


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


OUT32 = os.path.join(HERE, "libdmenv32.so")
# Backend options (round 4, A/B in profiles/r04_ab_kernel_variants.md).  The step kernels run ONE wave per SIMD with the whole register file: the default
# scheduling strategy (max-occupancy: keep register pressure low) buys nothing there, scheduling for instruction-level parallelism hides more of a lone
# wave's LDS / f64 latencies (+1.0 .. +1.6 % env-steps/s); register-class priority in the greedy allocator +0.4 %.  Neither touches floating-point
# semantics: results are bit-identical.
BACKEND_FLAGS = ["-mllvm", "-amdgpu-sched-strategy=max-ilp", "-mllvm", "-greedy-regclass-priority-trumps-globalness=1"]


def _stale(out, srcs):
    return not os.path.exists(out) or any(os.path.getmtime(s) > os.path.getmtime(out) for s in srcs)


def build(force=False, verbose=False):
    """libdmenv.so (float64 arithmetic: the parity build) and libdmenv32.so (-DDM_REAL_FLOAT: the same kernels in float32, the
    `dtype 32` batch), compiled side by side.  Returns the path of libdmenv.so."""
    srcs = [os.path.join(HERE, s) for s in SOURCES] + [os.path.join(REPO, "include", "dmenv.h")]
    jobs = []
    for out, defs in ((OUT, []), (OUT32, ["-DDM_REAL_FLOAT"])):
        if not force and not _stale(out, srcs):
            continue
        cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value", "-Wno-shift-count-negative",
               "-Wno-implicit-const-int-float-conversion"] + BACKEND_FLAGS + defs + os.environ.get("DM_BUILD_DEFINES", "").split() + [
               "-I" + os.path.join(REPO, "include"), "-I" + HERE, os.path.join(HERE, "dmenv.hip"), "-o", out]
        if verbose:
            cmd.insert(-2, "-Rpass-analysis=kernel-resource-usage")
            print(" ".join(cmd))
        jobs.append((cmd, subprocess.Popen(cmd)))
    for cmd, pr in jobs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
