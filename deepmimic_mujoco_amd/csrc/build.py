"""Build libdmenv.so (HIP, gfx950) in-tree with hipcc.  Usage: python -m deepmimic_mujoco_amd.csrc.build [--force]"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
SOURCES = ["dmenv.hip", "kernels_packed.hip", "kernels_rollout.hip", "kernels.h", "env_kernel.h", "env_step.h", "model_host.h", "topology.h", "wave.h", "policy_kernel.h", "vf_kernel.h", "pg_kernel.h", "slot_kernel.h", "slot_step.h"]
OUT = os.path.join(HERE, "libdmenv.so")
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-shift-count-negative", "-Wno-implicit-const-int-float-conversion"]
# Two translation units, each with the backend options its kernels want (round 4, A/B in profiles/r04_ab_kernel_variants.md):
#   kernels_packed.hip — four environments per wavefront: ONE wave per SIMD with the whole register file.  The default scheduling strategy (max-occupancy: keep
#                        register pressure low) buys nothing there; scheduling for instruction-level parallelism hides more of a lone wave's LDS / f64 latencies
#                        (+1.0 .. +1.6 % env-steps/s), register-class priority in the greedy allocator +0.4 %.
#   dmenv.hip          — the one-env step kernels (two waves per SIMD at 256 registers: 2 % SLOWER under max-ilp), everything else, the C ABI: defaults.
# Neither option touches floating-point semantics: results are bit-identical.
#                        -enable-ipra (round 6): inter-procedural register allocation — the horizon launch's called step bodies (internal functions, slot_step.h
#                        DM_CALL_SLOT) are compiled without callee-saved registers: no 337-register save / restore per call, +3.7 % (profiles/r06_ab_kernel_variants.md §4).
PACKED_FLAGS = ["-mllvm", "-amdgpu-sched-strategy=max-ilp", "-mllvm", "-greedy-regclass-priority-trumps-globalness=1", "-mllvm", "-enable-ipra"]
if os.environ.get("DM_PACKED_FLAGS") is not None:      # experiments: replace the packed unit's backend options altogether (tools/build_variant.sh)
    PACKED_FLAGS = os.environ["DM_PACKED_FLAGS"].split()
#   kernels_rollout.hip — k_rollout_packed and its called step bodies: the packed options + the IR-level load-store vectoriser off (round 6: +0.8 .. +1.4 % on the
#                        horizon launch, -1 % on the per-step packed kernel, hence a unit of its own).
#                        -amdgpu-use-amdgpu-trackers: the backend's own register-pressure trackers in the scheduler: +0.5 .. +1.3 % there (section 17 of the same file).
ROLLOUT_FLAGS = PACKED_FLAGS + ["-mllvm", "-amdgpu-load-store-vectorizer=false", "-mllvm", "-amdgpu-use-amdgpu-trackers=true"]
if os.environ.get("DM_ROLLOUT_FLAGS") is not None:
    ROLLOUT_FLAGS = os.environ["DM_ROLLOUT_FLAGS"].split()
UNITS = [("dmenv.hip", []), ("kernels_packed.hip", PACKED_FLAGS), ("kernels_rollout.hip", ROLLOUT_FLAGS)]


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


OUT32 = os.path.join(HERE, "libdmenv32.so")

def _stale(out, srcs):
    return not os.path.exists(out) or any(os.path.getmtime(s) > os.path.getmtime(out) for s in srcs)


def build_one(out, defs=(), extra=(), verbose=False, objdir=None):
    """Compile the translation units side by side and link them into `out`.  defs: extra -D... for every unit; extra: extra compiler options for every unit."""
    objdir = objdir or os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    tag = os.path.splitext(os.path.basename(out))[0]
    jobs, objs = [], []
    for src, flags in UNITS:
        obj = os.path.join(objdir, "%s_%s.o" % (tag, os.path.splitext(src)[0]))
        flags = list(flags) + os.environ.get("DM_FLAGS_" + os.path.splitext(src)[0].upper(), "").split()      # experiments: DM_FLAGS_DMENV / DM_FLAGS_KERNELS_PACKED
        cmd = [hipcc()] + COMMON + list(flags) + list(defs) + list(extra) + ["-I" + os.path.join(REPO, "include"), "-I" + HERE, "-c", os.path.join(HERE, src), "-o", obj]
        if verbose:
            cmd.insert(-4, "-Rpass-analysis=kernel-resource-usage")
            print(" ".join(cmd))
        jobs.append((cmd, subprocess.Popen(cmd))); objs.append(obj)
    for cmd, pr in jobs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    link = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out]
    subprocess.check_call(link)
    return out


def build(force=False, verbose=False):
    """libdmenv.so (float64 arithmetic: the parity build) and libdmenv32.so (-DDM_REAL_FLOAT: the same kernels in float32, the
    `dtype 32` batch), compiled side by side.  Returns the path of libdmenv.so."""
    import threading
    srcs = [os.path.join(HERE, s) for s in SOURCES] + [os.path.join(REPO, "include", "dmenv.h"), os.path.abspath(__file__)]      # (this file: the units' backend options)
    extra = os.environ.get("DM_BUILD_DEFINES", "").split()
    todo = [(out, defs) for out, defs in ((OUT, []), (OUT32, ["-DDM_REAL_FLOAT"])) if force or _stale(out, srcs)]
    errs = []

    def run(out, defs):
        try:
            build_one(out, defs, extra, verbose)
        except Exception as e:       # noqa: BLE001
            errs.append(e)
    ths = [threading.Thread(target=run, args=t) for t in todo]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    if errs:
        raise errs[0]
    return OUT


if __name__ == "__main__":
    if "--out" in sys.argv:        # python build.py --out build_ab/X.so [compiler options ...]: one float64 library with extra options (A/B builds: tools/build_variant.sh)
        i = sys.argv.index("--out")
        print(build_one(os.path.abspath(sys.argv[i + 1]), extra=sys.argv[i + 2:], objdir=os.path.join(os.path.dirname(os.path.abspath(sys.argv[i + 1])), "obj")))
    else:
        print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
