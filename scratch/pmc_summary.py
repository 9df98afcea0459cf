"""Per-kernel means of every counter in a rocprofv3 --pmc output dir -> <dir>/pmc_by_kernel.csv (and print)."""
import csv, glob, os, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        a = agg[r["Kernel_Name"][:100]][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
names = sorted({c for k in agg for c in agg[k]})
rows = []
for k, d in agg.items():
    rows.append([k, max(v[0] for v in d.values())] + [d[c][1] / d[c][0] if c in d else "" for c in names])
rows.sort(key=lambda r: -float(r[2] or 0))
with open(os.path.join(out, "pmc_by_kernel.csv"), "w") as f:
    w = csv.writer(f); w.writerow(["kernel", "launches"] + names); w.writerows(rows)
print(",".join(["kernel", "launches"] + names))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    print(r[0][:70], r[1], *[f"{x:.3g}" if x != "" else "-" for x in r[2:]])
