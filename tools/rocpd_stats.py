#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME` writes
NAME_results.db on ROCm 7.2) as the per-kernel table `--stats` would print: calls, total / average / min /
max duration and share of GPU time.  Usage: tools/rocpd_stats.py results.db [out.md]"""
import sqlite3
import sys


def main(db_path, out_path=None):
    cur = sqlite3.connect(db_path).cursor()
    rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels group by name order by sum(end-start) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for name, n, tot, avg, mn, mx in rows:
        lines.append("| `%s` | %d | %.3f | %.2f | %.2f | %.2f | %.1f |" % (name[:70], n, tot / 1e6, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total))
    # real (non-empty) launches only: the engine over-launches a little past the end of a solve and those
    # kernels exit at once; a 1.5 us cut separates them cleanly from real work on the large LP
    for kern in ("k_update", "k_select", "k_pivot_fused"):
        r = cur.execute("select count(*), avg(end-start) from kernels where name like ? and (end-start) > 3000", (kern + "%",)).fetchone()
        if r and r[0]:
            lines.append("")
            lines.append("`%s` launches longer than 3 us (real work): %d, average %.2f us" % (kern, r[0], r[1] / 1e3))
    text = "\n".join(lines) + "\n"
    if out_path:
        with open(out_path, "w") as fh:
            fh.write(text)
    print(text)


if __name__ == "__main__":
    main(*sys.argv[1:3])
