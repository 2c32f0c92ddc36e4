#!/usr/bin/env python3
"""Busy / idle analysis of a rocprofv3 kernel trace (.db): over the last `window_ms` of the trace, the union of kernel
intervals (any queue) against wall time, the concurrency histogram, and the kernels that follow the longest idle gaps."""
import sqlite3, sys, collections

def main(path, window_ms=60.0, anchor=None):
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    t_end = max(r[2] for r in rows if anchor is None or anchor in r[0])       # window ends with the last `anchor` kernel
    rows = [r for r in rows if r[1] < t_end]
    t0 = t_end - int(window_ms * 1e6)
    rows = [r for r in rows if r[2] > t0]
    ev = []
    for n, s, e in rows:
        ev.append((max(s, t0), 1)); ev.append((e, -1))
    ev.sort()
    depth = 0; last = t0; hist = collections.Counter()
    for t, d in ev:
        hist[depth] += t - last; last = t; depth += d
    wall = t_end - t0
    print(f"window {wall/1e6:.2f} ms, {len(rows)} kernels")
    for k in sorted(hist): print(f"  concurrency {k}: {hist[k]/1e6:8.3f} ms  {100*hist[k]/wall:5.1f} %")
    # gaps: idle intervals (depth 0) and what starts after them
    gaps = []
    depth = 0; last_end = t0
    cur_end = t0
    for n, s, e in rows:
        if s > cur_end: gaps.append((s - cur_end, n))
        cur_end = max(cur_end, e)
    gaps.sort(reverse=True)
    print("longest idle gaps (us) and the kernel that ends them:")
    for g, n in gaps[:12]: print(f"  {g/1e3:8.1f}  {n[:90]}")
    agg = collections.Counter()
    for g, n in gaps: agg[n[:60]] += g
    print("idle time by following kernel (us):")
    for n, g in agg.most_common(10): print(f"  {g/1e3:8.1f}  {n}")

if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 60.0, sys.argv[3] if len(sys.argv) > 3 else None)
