"""Build libdmenv.so (HIP, gfx950) in-tree with hipcc.  Usage: python -m deepmimic_mujoco_amd.csrc.build [--force]"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
SOURCES = ["dmenv.hip", "env_kernel.h", "env_step.h", "model_host.h", "topology.h", "wave.h", "policy_kernel.h"]
OUT = os.path.join(HERE, "libdmenv.so")


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def build(force=False, verbose=False):
    srcs = [os.path.join(HERE, s) for s in SOURCES] + [os.path.join(REPO, "include", "dmenv.h")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(s) <= os.path.getmtime(OUT) for s in srcs):
        return OUT
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value",
           "-I" + os.path.join(REPO, "include"), "-I" + HERE, os.path.join(HERE, "dmenv.hip"), "-o", OUT]
    if verbose:
        cmd.insert(-2, "-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
