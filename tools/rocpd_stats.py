#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd .db (kernel-trace) into the familiar --stats table.

rocprofv3 in ROCm 7.2 writes a rocpd sqlite database by default; this prints (and
optionally saves) per-kernel count / total / average / min / max duration and the share
of GPU time, plus VGPR/SGPR/LDS per kernel when recorded.
usage: tools/rocpd_stats.py results.db [out.md]
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(
        "select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
        "from kernels group by %s order by 3 desc" % (name_col, name_col)).fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for n, c, tot, avg, mn, mx in rows:
        short = n if len(n) < 90 else n[:87] + "..."
        lines.append("| %s | %d | %.3f | %.2f | %.2f | %.2f | %.1f |" % (short, c, tot / 1e6, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total))
    extra = [c for c in ("vgpr_count", "accum_vgpr_count", "sgpr_count", "lds_size", "scratch_size", "grid_x", "workgroup_x") if c in cols]
    if extra:
        lines += ["", "| kernel | " + " | ".join(extra) + " |", "|---|" + "---|" * len(extra)]
        for r in cur.execute("select %s, %s from kernels group by %s" % (name_col, ", ".join("max(%s)" % c for c in extra), name_col)):
            short = r[0] if len(r[0]) < 90 else r[0][:87] + "..."
            lines.append("| %s | %s |" % (short, " | ".join(str(x) for x in r[1:])))
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
