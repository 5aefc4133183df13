"""Kernels between two named launches of a rocprofv3 kernel trace:  python tools/gpu_span.py <trace.db> <from substring> <to substring> [occurrence]
Prints start offset (us), duration (us) and name of every kernel from the `occurrence`-th launch matching <from> to the next one matching <to>."""
import sqlite3
import sys


def main():
    db, a, b = sys.argv[1], sys.argv[2], sys.argv[3]
    occ = int(sys.argv[4]) if len(sys.argv) > 4 else 10
    rows = sqlite3.connect(db).cursor().execute("select name, start, end from kernels order by start").fetchall()
    idx = [i for i, r in enumerate(rows) if a in r[0]]
    i0 = idx[min(occ, len(idx) - 1)]
    t0 = rows[i0][1]
    for name, s, e in rows[i0:]:
        print("%9.1f  %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, name.split("(")[0][-70:]))
        if b in name and s > t0:
            break


if __name__ == "__main__":
    main()
