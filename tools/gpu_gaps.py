"""GPU idle gaps of a rocprofv3 kernel trace (rocpd sqlite):  python tools/gpu_gaps.py <trace.db> [<anchor kernel substring>]
Splits the trace at each launch of the anchor kernel (default k_rollout_packed) and prints, for the median period: busy time (union of kernel
intervals), idle time, and the largest idle gaps with the kernels on either side."""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    anchor = sys.argv[2] if len(sys.argv) > 2 else "k_rollout_packed"
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    starts = [i for i, r in enumerate(rows) if anchor in r[0]]
    if len(starts) < 4:
        print("too few anchor launches"); return
    k = starts[len(starts) // 2]; k2 = starts[len(starts) // 2 + 1]
    per = rows[k:k2 + 1]
    t0, t1 = per[0][1], per[-1][1]
    print("period %.2f ms (%d launches)" % ((t1 - t0) / 1e6, len(per) - 1))
    busy_end = per[0][2]; busy = per[0][2] - per[0][1]; gaps = []
    last = per[0][0]
    for name, s, e in per[1:]:
        if s > busy_end:
            gaps.append((s - busy_end, last, name, (busy_end - t0) / 1e6))
            if name is not per[-1][0] or True:
                pass
            busy += (e - s) if name is not None else 0
        else:
            busy += max(0, e - busy_end)
        if e > busy_end:
            busy_end = e; last = name
    idle = sum(g[0] for g in gaps)
    print("idle %.2f ms in %d gaps;  gaps > 20 us:" % (idle / 1e6, len(gaps)))
    acc = {}
    for g in gaps:
        key = (g[1].split("(")[0][-40:], g[2].split("(")[0][-40:])
        a = acc.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += g[0] / 1e3
    for key, a in sorted(acc.items(), key=lambda x: -x[1][1])[:25]:
        print("  %8.1f us  x%-4d  after %-40s before %s" % (a[1], a[0], key[0], key[1]))
    print("timeline of gaps > 150 us:")
    for g in gaps:
        if g[0] > 150e3:
            print("  at %6.2f ms: %7.1f us  after %s  before %s" % (g[3], g[0] / 1e3, g[1].split("(")[0][-40:], g[2].split("(")[0][-40:]))


if __name__ == "__main__":
    main()
