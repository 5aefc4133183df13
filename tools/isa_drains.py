"""Where a lone wave waits: full drains of the LDS / memory counters per barrier-delimited section of the packed step's listing.

One wave per SIMD has no other wave to switch to, so every `s_waitcnt lgkmcnt(0)` behind an LDS read is a round trip the wave sits out (~130 cycles), and every
`s_waitcnt vmcnt(0)` behind a global load a trip to the L2 (~500+).  The instruction count does not show them (profiles/r05_ab_kernel_variants.md section 7: the
mass-matrix entries took 11 cycles per instruction until their operands were requested in groups).  This tool lists, for each stage (DM_MARK) of
slot_env_step_call<double, 32> and each section between two wave barriers inside it: instructions, LDS reads, global / scratch loads, FULL drains of lgkmcnt and of
vmcnt, partial waits.  Sections with many drains per read are the candidates for "operands first, scheduling fence, then the arithmetic".
Static counts: a section inside a divergent branch or a loop runs as often as its lanes / trips say.
usage: python tools/isa_drains.py [out.md]      (DM_BUILD_DEFINES / DM_ISA_OUT as for tools/isa_mix_packed.py)"""
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

SYM = "_ZN2dmL18slot_env_step_callIdLi32E"


def main():
    out_md = sys.argv[1] if len(sys.argv) > 1 else None
    s_path = os.environ.get("DM_ISA_OUT", os.path.join(tempfile.gettempdir(), "dmenv_packed_isa.s"))
    cmd = [B.hipcc()] + [f for f in B.COMMON if f != "-fPIC"] + B.ROLLOUT_FLAGS + os.environ.get("DM_BUILD_DEFINES", "").split() + [
        "-I" + os.path.join(ROOT, "include"), "-I" + CS, "-S", "--cuda-device-only", os.path.join(CS, "kernels_rollout.hip"), "-o", s_path]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    rows = []
    on = False
    stage, sec = "entry", 0
    c = collections.Counter()

    def flush():
        if c["n"]:
            rows.append((stage, sec, dict(c)))

    for line in open(s_path):
        if line.startswith(SYM):
            on = True
            continue
        if not on:
            continue
        if line.startswith(".Lfunc_end"):
            break
        m = re.search(r"; DM_MARK (\S+)", line)
        if m:
            flush(); c = collections.Counter(); stage, sec = m.group(1), 0
            continue
        if "; wave barrier" in line:
            flush(); c = collections.Counter(); sec += 1
            continue
        t = line.strip()
        if not t or t[0] in ";." or t.endswith(":"):
            continue
        op = t.split()[0]
        c["n"] += 1
        if op == "s_waitcnt":
            if "lgkmcnt(0)" in t:
                c["lds_drain"] += 1
            elif "lgkmcnt" in t:
                c["lds_partial"] += 1
            if "vmcnt(0)" in t:
                c["vm_drain"] += 1
            elif "vmcnt" in t:
                c["vm_partial"] += 1
        elif op.startswith("ds_read"):
            c["lds_read"] += 1
        elif op.startswith("ds_"):
            c["lds_write"] += 1
        elif op.startswith(("global_load", "scratch_load", "flat_load")):
            c["mem_load"] += 1
    flush()
    lines = ["# Full drains per section of the packed step (`tools/isa_drains.py`; lean instantiation `slot_env_step_call<double, 32>`, product flags)", "",
             __doc__.split("usage:")[0].strip(), "",
             "| stage | section | instructions | LDS reads | LDS writes / atomics | global + scratch loads | full lgkmcnt drains | partial | full vmcnt drains | partial |", "|---|---|---|---|---|---|---|---|---|---|"]
    tot = collections.Counter()
    per_stage = collections.OrderedDict()
    for st, k, d in rows:
        lines.append("| %s | %d | %d | %d | %d | %d | %d | %d | %d | %d |" % (st, k, d.get("n", 0), d.get("lds_read", 0), d.get("lds_write", 0), d.get("mem_load", 0),
                                                                           d.get("lds_drain", 0), d.get("lds_partial", 0), d.get("vm_drain", 0), d.get("vm_partial", 0)))
        tot.update(d); per_stage.setdefault(st, collections.Counter()).update(d)
    lines += ["", "## per stage", "", "| stage | instructions | LDS reads | global + scratch loads | full lgkmcnt drains | full vmcnt drains |", "|---|---|---|---|---|---|"]
    for st, d in per_stage.items():
        lines.append("| %s | %d | %d | %d | %d | %d |" % (st, d["n"], d["lds_read"], d["mem_load"], d["lds_drain"], d["vm_drain"]))
    lines.append("| **total** | %d | %d | %d | %d | %d |" % (tot["n"], tot["lds_read"], tot["mem_load"], tot["lds_drain"], tot["vm_drain"]))
    txt = "\n".join(lines) + "\n"
    if out_md:
        open(out_md, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
