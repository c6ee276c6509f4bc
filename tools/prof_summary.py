#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace results .db (rocpd sqlite) into a per-kernel stats table
(the same content as rocprofv3's --stats CSV): calls, total / avg / min / max duration, share."""
import re
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = db.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                      f"from kernels group by {name_col} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = [f"{'kernel':<88} {'calls':>7} {'total_ms':>10} {'avg_us':>9} {'min_us':>9} {'max_us':>9} {'pct':>6}"]
    for n, c, s, a, mn, mx in rows:
        n = re.sub(r"\(.*$", "", n)[:88]
        lines.append(f"{n:<88} {c:>7} {s/1e6:>10.3f} {a/1e3:>9.2f} {mn/1e3:>9.2f} {mx/1e3:>9.2f} {100*s/total:>6.2f}")
    text = "\n".join(lines)
    print(text)
    if out:
        open(out, "w").write(text + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
