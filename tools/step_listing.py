#!/usr/bin/env python
"""Ordered kernel list of ONE steady-state training step from a rocprofv3 --kernel-trace .db: start offset, duration,
gap to the latest earlier end, name (template arguments kept, argument lists cut).  usage: step_listing.py results.db [--step 10]"""
import argparse
import re
import sqlite3

ap = argparse.ArgumentParser()
ap.add_argument("db")
ap.add_argument("--mark", default="adam_kernel")
ap.add_argument("--step", type=int, default=10)
a = ap.parse_args()
db = sqlite3.connect(a.db)
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
extra = [c for c in ("grid_x", "grid_size_x", "workgroup_size_x") if c in cols]
rows = db.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
rows = [(re.sub(r"\(.*$", "", n), s, e) for n, s, e in rows]
marks = [i for i, (n, s, e) in enumerate(rows) if a.mark in n]
i0, i1 = marks[a.step] + 1, marks[a.step + 1] + 1
t0 = rows[i0][1]
latest = t0
tot = 0
for n, s, e in rows[i0:i1]:
    gap = max(0, s - latest)
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap / 1e3:6.1f}  {n[:150]}")
    latest = max(latest, e)
    tot += e - s
print(f"# {i1 - i0} launches, kernel time {tot / 1e6:.3f} ms, span {(latest - t0) / 1e6:.3f} ms")
