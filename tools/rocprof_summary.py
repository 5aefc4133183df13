"""Summarise rocprofv3 (rocpd sqlite) outputs into markdown: kernel-trace stats + PMC counters per kernel.
usage: python tools/rocprof_summary.py <out.md> <title> <trace.db> [<pmc.db> ...]"""
import sqlite3
import sys


def main():
    out, title, trace, pmcs = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4:]
    lines = ["# " + title, ""]
    cur = sqlite3.connect(trace).cursor()
    rows = cur.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start), max(vgpr_count), "
                       "max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size) from kernels group by name order by 6 desc").fetchall()
    tot = sum(r[5] for r in rows) or 1
    lines += ["## Kernel trace (`rocprofv3 --kernel-trace --stats`)", "",
              "| kernel | calls | avg us | min us | max us | total ms | % | VGPR | AGPR | SGPR | LDS B | scratch B/lane |", "|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for r in rows[:int(__import__("os").environ.get("ROWS", "8"))]:
        lines.append("| %s | %d | %.1f | %.1f | %.1f | %.2f | %.1f | %s | %s | %s | %s | %s |" % (
            r[0].split("(")[0][:48], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e6, 100 * r[5] / tot, r[6], r[7], r[8], r[9], r[10]))
    lines += ["", "## PMC (one counter group per pass; averages per launch)", "", "| kernel | counter | avg per launch | launches |", "|---|---|---|---|"]
    for db in pmcs:
        cur = sqlite3.connect(db).cursor()
        for r in cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where (kernel_name like 'k_step%' or kernel_name like 'k_rollout%') "
                             "group by kernel_name, counter_name").fetchall():
            lines.append("| %s | %s | %.1f | %d |" % (r[0].split("(")[0], r[1], r[2], r[3]))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
