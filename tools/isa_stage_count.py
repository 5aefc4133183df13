"""Static instruction mix per stage of k_step_narrow: splits the device assembly on the DM_MARK comments.
usage: python tools/isa_stage_count.py [kernel-symbol-prefix]   (straight-line/unrolled code only: loops count once)"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "deepmimic_mujoco_amd", "csrc")


def main():
    sym = sys.argv[1] if len(sys.argv) > 1 else "_Z13k_step_narrow"
    out = os.path.join(tempfile.gettempdir(), "dmenv_isa.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-I" + os.path.join(ROOT, "include"),
                           "-I" + CS, "-S", "--cuda-device-only", os.path.join(CS, "dmenv.hip"), "-o", out], stderr=subprocess.DEVNULL)
    on = False
    stage = "prologue"
    counts = collections.OrderedDict()
    for line in open(out):
        if line.startswith(sym):
            on = True
            continue
        if not on:
            continue
        m = re.search(r"; DM_MARK (\S+)", line)
        if m:
            stage = m.group(1)
            continue
        t = line.strip()
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        op = t.split()[0]
        c = counts.setdefault(stage, collections.Counter())
        kind = ("valu_f64" if re.match(r"v_.*_f64", op) else "valu" if op.startswith("v_") else "salu" if op.startswith("s_") and not op.startswith("s_waitcnt") and not op.startswith("s_nop")
                else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "flat_", "buffer_", "scratch_")) else "wait" if op.startswith(("s_waitcnt", "s_nop")) else "other")
        c[kind] += 1
        if op == "s_endpgm":
            break
    print("%-22s %8s %8s %8s %8s %8s %8s" % ("stage (static count)", "valu_f64", "valu", "salu", "lds", "vmem", "wait"))
    tot = collections.Counter()
    for st, c in counts.items():
        print("%-22s %8d %8d %8d %8d %8d %8d" % (st, c["valu_f64"], c["valu"], c["salu"], c["lds"], c["vmem"], c["wait"]))
        tot.update(c)
    print("%-22s %8d %8d %8d %8d %8d %8d" % ("total", tot["valu_f64"], tot["valu"], tot["salu"], tot["lds"], tot["vmem"], tot["wait"]))


if __name__ == "__main__":
    main()
