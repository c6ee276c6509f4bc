#!/usr/bin/env python
"""Average rocprofv3 PMC counters per (kernel, grid) from p_counter_collection.csv files."""
import csv, sys, collections, re
def main(paths):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in paths:
        for row in csv.DictReader(open(path)):
            k = (re.sub(r"\(.*$", "", row["Kernel_Name"])[:60], row.get("Grid_Size", ""))
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, d in sorted(agg.items()):
        print(k)
        for c, v in sorted(d.items()):
            print(f"    {c:28s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
if __name__ == "__main__":
    main(sys.argv[1:])
