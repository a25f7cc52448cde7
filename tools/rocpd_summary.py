#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average duration.
usage: tools/rocpd_summary.py <results.db> [--grids]   (writes the table to stdout; keep it under profiles/)"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                       "group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print(f"{'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>9} {'max_us':>9} {'pct':>6}  kernel")
    for name, n, s, a, mn, mx in rows:
        print(f"{n:7d} {s / 1e6:10.3f} {a / 1e3:10.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} {100.0 * s / tot:6.2f}  {name[:150]}")
    print(f"total kernel time {tot / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
    if "--grids" in sys.argv:
        print("\nper (kernel, grid) breakdown:")
        for name, gx, gy, n, s, a in cur.execute(
                "select name, grid_x, grid_y, count(*), sum(duration), avg(duration) from kernels group by name, grid_x, grid_y "
                "order by sum(duration) desc limit 60"):
            print(f"{n:7d} {s / 1e6:10.3f} {a / 1e3:10.2f}  grid=({gx},{gy})  {name[:110]}")




def pmc_summary(path):
    """Per-kernel averages of every PMC counter in a rocprofv3 --pmc results db."""
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute("select kernel_name, counter_name, count(*), sum(value), avg(value), avg(duration) from counters_collection "
                       "group by kernel_name, counter_name order by sum(value) desc").fetchall()
    print(f"{'launches':>8} {'sum':>16} {'avg/launch':>14} {'avg_us':>9}  counter  kernel")
    for kn, cn, n, sm, av, du in rows:
        if sm and sm > 0:
            print(f"{n:8d} {sm:16.1f} {av:14.2f} {du / 1e3:9.2f}  {cn:28s} {kn[:110]}")
    return rows


def pmc_by_grid(path, top=40):
    """Per (kernel, grid) averages of every counter: attributes a variant's traffic to the individual launch shapes of a step
    (`--pmc-grids`; a launch shape = one convolution of the model here: same kernel + same grid)."""
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute("select kernel_name, grid_size_x, grid_size_y, counter_name, count(*), sum(value), avg(value), avg(duration) from counters_collection "
                       "group by kernel_name, grid_size_x, grid_size_y, counter_name order by sum(value) desc").fetchall()
    print(f"{'launches':>8} {'sum':>16} {'avg/launch':>14} {'avg_us':>9}  counter  grid  kernel")
    for kn, gx, gy, cn, n, sm, av, du in rows[:top]:
        if sm and sm > 0:
            print(f"{n:8d} {sm:16.1f} {av:14.2f} {du / 1e3:9.2f}  {cn:14s} ({gx},{gy})  {kn[:100]}")


def timeline(path, n):
    """The last n dispatches in start order: offset from the first of them, duration, gap to the latest end before it, queue."""
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
    q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    rows = cur.execute(f"select name, start, end, {q}, grid_x from kernels order by start desc limit {int(n)}").fetchall()[::-1]
    t0 = rows[0][1]; last_end = rows[0][1]
    print(f"{'t_us':>9} {'dur_us':>8} {'gap_us':>8} {'q':>3} {'grid':>8}  kernel")
    for name, st, en, qi, gx in rows:
        print(f"{(st - t0) / 1e3:9.1f} {(en - st) / 1e3:8.1f} {(st - last_end) / 1e3:8.1f} {qi!s:>3} {gx:8d}  {name[:120]}")
        last_end = max(last_end, en)
    busy = sum(r[2] - r[1] for r in rows)
    print(f"span {(max(r[2] for r in rows) - t0) / 1e3:.1f} us, sum of durations {busy / 1e3:.1f} us, {len(rows)} dispatches")


if __name__ == "__main__":
    if "--timeline" in sys.argv:
        timeline(sys.argv[1], sys.argv[sys.argv.index("--timeline") + 1])
    elif "--pmc-grids" in sys.argv:
        pmc_by_grid(sys.argv[1])
    elif "--pmc" in sys.argv:
        pmc_summary(sys.argv[1])
    else:
        main()
