#!/usr/bin/env python
"""Per-launch HBM-side traffic of a kernel family from the separate FETCH_SIZE / WRITE_SIZE PMC passes
(tools/pmc_run.sh).  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE tallies 128-B requests at
64 B for wide coalesced streams -> doubled; WRITE_SIZE is used as reported (calibrated here: the FFN conv launch
writes 84672 KiB = its algorithmic 21168 x 1024 fp32 output)."""
import csv, json, re, sys

def mean_counter(path, counter, pattern):
    tot, n = 0.0, 0
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] == counter and re.search(pattern, row["Kernel_Name"]):
            tot += float(row["Counter_Value"]); n += 1
    return (tot / n if n else 0.0), n

def main(outdir, pattern, label):
    f, n = mean_counter(f"{outdir}/fetch/p_counter_collection.csv", "FETCH_SIZE", pattern)
    w, _ = mean_counter(f"{outdir}/write/p_counter_collection.csv", "WRITE_SIZE", pattern)
    h, _ = mean_counter(f"{outdir}/tcc/p_counter_collection.csv", "TCC_HIT_sum", pattern)
    m, _ = mean_counter(f"{outdir}/tcc/p_counter_collection.csv", "TCC_MISS_sum", pattern)
    print(json.dumps({"label": label, "kernel_regex": pattern, "launches_sampled": n,
                      "fetch_bytes_per_launch_corrected": round(2 * f * 1024), "write_bytes_per_launch": round(w * 1024),
                      "traffic_bytes_per_launch": round((2 * f + w) * 1024),
                      "l2_hit_rate": round(h / (h + m), 4) if h + m else None}))

if __name__ == "__main__":
    main(*sys.argv[1:4])
