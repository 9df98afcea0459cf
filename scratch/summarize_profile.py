"""Condense rocprofv3 outputs of scratch/profile_round.sh: per-kernel PMC means -> pmc_summary.csv."""
import csv, glob, os, sys, collections
out = sys.argv[1]
rows = []
for tag, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    files = glob.glob(os.path.join(out, tag, "**", "*counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            key = (r["Kernel_Name"][:110], r.get("Grid_Size", r.get("Grid_Size_X", "")))
            a = agg[key]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    for (k, g), (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        rows.append((counter, k, g, n, tot / n))
with open(os.path.join(out, "pmc_summary.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["counter", "kernel", "grid_threads", "launches", "mean_value_KB"])
    w.writerows(rows)
print("pmc rows", len(rows))
