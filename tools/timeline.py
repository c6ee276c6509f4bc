#!/usr/bin/env python
"""Timeline of a rocprofv3 --kernel-trace results .db (rocpd sqlite): how much of the wall time of the steady-state steps has
no kernel running (launch gaps), one kernel, or two or more (side streams overlapping), and which kernels sit behind the
longest gaps.  The step boundary is the first launch of the kernel given by --mark (default: the optimiser's adam_kernel).

usage: tools/timeline.py results.db [--mark adam_kernel] [--skip 6] [--top 25]"""
import argparse
import re
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--mark", default="adam_kernel")
    ap.add_argument("--skip", type=int, default=6, help="steps to skip at the start (warm-up, capture)")
    ap.add_argument("--top", type=int, default=25)
    a = ap.parse_args()
    db = sqlite3.connect(a.db)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = db.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
    rows = [(re.sub(r"\(.*$", "", n), s, e) for n, s, e in rows]
    marks = [e for n, s, e in rows if a.mark in n]
    # one mark per step: merge marks closer than 1 ms
    steps = []
    for m in marks:
        if not steps or m - steps[-1] > 1_000_000:
            steps.append(m)
        else:
            steps[-1] = m
    if len(steps) < a.skip + 3:
        print(f"only {len(steps)} steps found"); return
    t0, t1 = steps[a.skip], steps[-1]
    nsteps = len(steps) - 1 - a.skip
    sel = [(n, s, e) for n, s, e in rows if s >= t0 and e <= t1]
    ev = []
    for n, s, e in sel:
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    depth, last = 0, t0
    hist = {0: 0, 1: 0, 2: 0}
    for t, d in ev:
        hist[min(depth, 2)] += t - last
        last = t
        depth += d
    hist[min(depth, 2)] += t1 - last
    wall = t1 - t0
    print(f"{nsteps} steps, {wall / nsteps / 1e6:.3f} ms/step wall, {len(sel) / nsteps:.0f} launches/step, "
          f"kernel time {sum(e - s for _, s, e in sel) / nsteps / 1e6:.3f} ms/step")
    for k, lab in ((0, "idle (no kernel)"), (1, "one kernel"), (2, "two or more")):
        print(f"  {lab:<18} {hist[k] / nsteps / 1e6:8.3f} ms/step  {100 * hist[k] / wall:5.1f} %")
    # gaps: idle interval before each kernel start (time since the latest end of everything earlier)
    gaps = {}
    latest_end = t0
    for n, s, e in sel:
        if s > latest_end:
            g = gaps.setdefault(n, [0, 0])
            g[0] += s - latest_end; g[1] += 1
        latest_end = max(latest_end, e)
    print("idle time in front of (kernel that ends the gap):")
    for n, (t, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[: a.top]:
        print(f"  {t / nsteps / 1e3:8.1f} us/step {c / nsteps:6.1f} gaps/step {t / c / 1e3:7.2f} us avg  {n[:90]}")


if __name__ == "__main__":
    main()
