#!/usr/bin/env python3
"""Summarise a rocprofv3 results .db (kernel trace) into a small text table: per-kernel calls / total / avg / %."""
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                      "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = [f"# rocprofv3 --kernel-trace --stats summary of {path}",
             f"# total kernel time {tot/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches",
             f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'%':>6s}"]
    for name, n, s, a, mn, mx in rows:
        short = name if len(name) <= 70 else name[:67] + "..."
        lines.append(f"{short:70s} {n:7d} {s/1e6:10.3f} {a/1e3:10.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100*s/tot:6.2f}")
    import os
    if os.environ.get("LONG_NAMES"):
        lines.append("")
        lines.append("# full names of the 60 largest entries")
        for name, n, s_, a, mn, mx in rows[:60]:
            lines.append(f"{s_/1e6:10.3f} ms {n:6d}  {name[:400]}")
    text = "\n".join(lines)
    print(text)
    if out:
        open(out, "w").write(text + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
