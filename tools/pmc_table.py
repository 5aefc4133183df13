#!/usr/bin/env python3
"""Fill the PMC table of a kernel summary (profiles/rNN_kstep*_summary.md) from the raw per-launch counter exports bench.py keeps with
DM_PROFILE_KEEP (profiles/raw/<prefix>_{sq,fetch,write,grbm}_counters.csv): average per launch and number of launches per (kernel, counter).
usage: python tools/pmc_table.py <summary.md> <raw prefix> [note]"""
import collections
import csv
import glob
import sys

md, prefix = sys.argv[1], sys.argv[2]
note = sys.argv[3] if len(sys.argv) > 3 else ""
acc = collections.OrderedDict()
for f in sorted(glob.glob(prefix + "_*_counters.csv")):
    for row in csv.reader(open(f)):
        if row[0] == "kernel":
            continue
        a = acc.setdefault((row[0], row[1]), [0.0, 0])
        a[0] += float(row[2]); a[1] += 1
lines = open(md).read().rstrip("\n").split("\n")
cut = [i for i, l in enumerate(lines) if l.startswith("## PMC")]
if cut:
    lines = lines[:cut[0]]
lines += ["## PMC (`rocprofv3 --kernel-trace --pmc ...`, one counter group per pass; averages per launch; raw per-launch values: `%s_*_counters.csv`)" % prefix, "",
          "| kernel | counter | avg per launch | launches |", "|---|---|---|---|"]
for (k, c), (s, n) in acc.items():
    if k.startswith("k_step") or k.startswith("k_order") or k.startswith("k_rollout"):
        lines.append("| %s | %s | %.1f | %d |" % (k, c, s / n, n))
if note:
    lines += ["", note]
open(md, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[-12:]))
