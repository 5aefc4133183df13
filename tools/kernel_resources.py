#!/usr/bin/env python3
"""Per-kernel resources of libdmenv.so as the CODE OBJECT states them (no GPU needed): the gfx950 ELF is cut out of the
library's clang offload bundle and its AMDGPU metadata note (llvm-readelf --notes) is listed per kernel — architectural and
accumulation VGPRs, SGPRs, spilled registers, LDS and scratch bytes, and the wave residency those imply.  This is the authoritative
figure wherever a profiler's trace database disagrees.   usage: python tools/kernel_resources.py [lib.so] [out.md]"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def code_objects(lib):
    """every gfx950 code object of the library: one clang offload bundle per translation unit (dmenv.hip, kernels_packed.hip)"""
    blob = open(lib, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out = []
    at = blob.find(magic)
    while at >= 0:
        n = struct.unpack_from("<Q", blob, at + len(magic))[0]
        p = at + len(magic) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "gfx950" in triple:
                out.append(blob[at + off:at + off + size])
        at = blob.find(magic, at + len(magic))
    if not out:
        raise SystemExit("no gfx950 code object in " + lib)
    return out


def demangle(sym):
    try:
        d = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip()
        return re.sub(r"^void ", "", d).split("(")[0]
    except Exception:
        return sym


def kernels(lib):
    txt = ""
    for co in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co); f.flush()
            txt += "\n" + subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True, check=True).stdout
    out = []
    for blk in re.split(r"\n\s+- \.agpr_count:", "\n" + txt)[1:]:
        blk = ".agpr_count:" + blk
        g = lambda k, d=None: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, d])[1]
        name = g("name")
        if not name:
            continue
        out.append(dict(name=demangle(name), vgpr=int(g("vgpr_count", 0)), agpr=int(g("agpr_count", 0)), sgpr=int(g("sgpr_count", 0)),
                        vspill=int(g("vgpr_spill_count", 0)), sspill=int(g("sgpr_spill_count", 0)), lds=int(g("group_segment_fixed_size", 0)),
                        scratch=int(g("private_segment_fixed_size", 0)), wg=int(g("max_flat_workgroup_size", 0))))
    return out


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "deepmimic_mujoco_amd", "csrc", "libdmenv.so")
    rows = sorted(kernels(lib), key=lambda r: -r["vgpr"])
    lines = ["| kernel | VGPR (unified total) | of which AGPR | SGPR | spilled VGPR | spilled SGPR | LDS B | scratch B/lane | waves/SIMD by registers | workgroups/CU by LDS |",
             "|---|---|---|---|---|---|---|---|---|---|"]
    for r in rows:
        tot = r["vgpr"]          # the note's .vgpr_count is the unified total: architectural + accumulation registers
        regs = 512 // max(8, (tot + 7) // 8 * 8) if tot else 8
        lds = (160 * 1024) // r["lds"] if r["lds"] else 32
        lines.append("| %s | %d | %d | %d | %d | %d | %d | %d | %d | %d |" % (r["name"][:40], r["vgpr"], r["agpr"], r["sgpr"], r["vspill"], r["sspill"], r["lds"],
                                                                              r["scratch"], min(8, regs), min(32, lds)))
    txt = "\n".join(lines)
    print(txt)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write("# Kernel resources of `%s` from its gfx950 code object (`tools/kernel_resources.py`, llvm-readelf --notes)\n\n%s\n"
                                     % (os.path.relpath(lib, ROOT), txt))


if __name__ == "__main__":
    main()
