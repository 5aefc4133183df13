"""Static instruction mix per stage of the packed step (the lean instantiation slot_env_step_call<double, 32> that k_rollout_packed calls, and the
three-set one <double, 40>): the device listing of csrc/kernels_rollout.hip — compiled with the product's backend options — split on the DM_MARK
comments, every instruction put into one bucket:
  f64 arith     v_*_f64 except moves / compares / DPP forms        dpp f64      v_fmac_f64_dpp / v_mov_b64_dpp (row broadcasts)
  dpp b32       v_mov_b32_dpp (lane permutations of sum16 etc.)    mov/sel      v_mov*, v_cndmask*, v_accvgpr_* (data movement inside the register file)
  int/addr      every other VALU instruction (index math, compares, conversions)
  lds           ds_*          vmem     global_* / flat_* / buffer_*          scratch  scratch_* (spills, callee-saved registers)
  salu          s_* except waits      wait     s_waitcnt / s_nop
Unrolled code counts once per copy, LOOPS COUNT ONCE (the PGS sweep loop, the narrow-phase trip loop): the table says what the code is made of, the
dynamic count per env-step (12.0 k VALU wave-instructions: bench.py roofline.pmc) says how often it runs.
usage: python tools/isa_mix_packed.py [out.md]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "deepmimic_mujoco_amd", "csrc")
sys.path.insert(0, ROOT)
from deepmimic_mujoco_amd.csrc import build as B  # noqa: E402

BUCKETS = ["f64 arith", "dpp f64", "dpp b32", "mov/sel", "int/addr", "lds", "vmem", "scratch", "salu", "wait"]


def bucket(t):
    op = t.split()[0]
    if op.startswith(("s_waitcnt", "s_nop")):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith(("global_", "flat_", "buffer_")):
        return "vmem"
    if op.startswith("v_"):
        dpp = " row_" in t or "quad_perm" in t or "row_newbcast" in t
        if dpp:
            return "dpp f64" if ("f64" in op or "b64" in op) else "dpp b32"
        if op.startswith(("v_mov", "v_cndmask", "v_accvgpr")):
            return "mov/sel"
        if re.match(r"v_(add|mul|fma|fmac|max|min|rcp|rsq|sqrt|div|ldexp|frexp|trig|fract|floor|ceil|rndne)\w*_f64", op):
            return "f64 arith"
        return "int/addr"
    return "salu"


def sections(path, sym):
    on = False
    stage = "entry (load, action, RK glue)"
    out = collections.OrderedDict()
    for line in open(path):
        if line.startswith(sym):
            on = True
            continue
        if not on:
            continue
        if line.startswith(".Lfunc_end"):
            break
        m = re.search(r"; DM_MARK (\S+)", line)
        if m:
            stage = m.group(1)
            continue
        t = line.strip()
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        out.setdefault(stage, collections.Counter())[bucket(t)] += 1
    return out


NAMES = [("slot_kinematics", "kinematics"), ("slot_bias", "bias forces"), ("slot_mass_factor", "mass matrix + factorisation"), ("slot_rows", "collision + rows"),
         ("slot_constraint_ns1_rows", "1 set: row build"), ("slot_constraint_ns1_8", "1 set: impedance, half solve, b"), ("slot_constraint_ns1_9", "1 set: A build"),
         ("slot_constraint_ns1_10", "1 set: warm start"), ("slot_constraint_ns1_11", "1 set: PGS (loop once)"), ("slot_constraint_ns1_12", "1 set: assembly + L solve"),
         ("slot_constraint_ns2_rows", "2 sets: row build"), ("slot_constraint_ns2_8", "2 sets: impedance, half solve, b"), ("slot_constraint_ns2_9", "2 sets: A build"),
         ("slot_constraint_ns2_10", "2 sets: park + warm start"), ("slot_constraint_ns2_11", "2 sets: PGS (loop once)"), ("slot_constraint_ns2_12", "2 sets: assembly + L solve"),
         ("slot_constraint_ns3_rows", "3 sets: row build"), ("slot_constraint_ns3_8", "3 sets: half solves, surplus rows, surplus blocks"), ("slot_constraint_ns3_9", "3 sets: A build"),
         ("slot_constraint_ns3_10", "3 sets: park + warm start"), ("slot_constraint_ns3_11", "3 sets: PGS (loop once)"), ("slot_constraint_ns3_12", "3 sets: assembly + L solve")]


def main():
    out_md = sys.argv[1] if len(sys.argv) > 1 else None
    s_path = os.environ.get("DM_ISA_OUT", os.path.join(tempfile.gettempdir(), "dmenv_packed_isa.s"))
    cmd = [B.hipcc()] + [f for f in B.COMMON if f != "-fPIC"] + B.ROLLOUT_FLAGS + os.environ.get("DM_BUILD_DEFINES", "").split() + ["-I" + os.path.join(ROOT, "include"), "-I" + CS, "-S", "--cuda-device-only",
                                                                                  os.path.join(CS, "kernels_rollout.hip"), "-o", s_path]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    lines = ["# Static instruction mix per stage of the packed step (`tools/isa_mix_packed.py`; gfx950 listing of `csrc/kernels_rollout.hip`, product flags)", "",
             __doc__.split("usage:")[0].strip(), ""]
    for sym, title in (("_ZN2dmL18slot_env_step_callIdLi32E", "lean instantiation `slot_env_step_call<double, 32>` (one and two row sets)"),
                       ("_ZN2dmL18slot_env_step_callIdLi40E", "three-set instantiation `slot_env_step_call<double, 40>`")):
        sec = sections(s_path, sym)
        lines += ["## " + title, "", "| stage | " + " | ".join(BUCKETS) + " | all | f64 arith share |", "|---|" + "---|" * (len(BUCKETS) + 2)]
        tot = collections.Counter()
        known = dict(NAMES)
        rest = collections.Counter()
        for st, c in sec.items():
            if st in known:
                n = sum(c.values())
                lines.append("| %s | " % known[st] + " | ".join(str(c[b]) for b in BUCKETS) + " | %d | %.0f %% |" % (n, 100.0 * (c["f64 arith"] + c["dpp f64"]) / max(1, n)))
            else:
                rest.update(c)
            tot.update(c)
        n = sum(rest.values())
        lines.append("| entry / RK glue / epilogue / callee-saved registers | " + " | ".join(str(rest[b]) for b in BUCKETS) + " | %d | %.0f %% |" % (n, 100.0 * (rest["f64 arith"] + rest["dpp f64"]) / max(1, n)))
        n = sum(tot.values())
        lines.append("| **total (static)** | " + " | ".join(str(tot[b]) for b in BUCKETS) + " | %d | %.0f %% |" % (n, 100.0 * (tot["f64 arith"] + tot["dpp f64"]) / max(1, n)))
        lines.append("")
    txt = "\n".join(lines) + "\n"
    if out_md:
        open(out_md, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
