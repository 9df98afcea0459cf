"""print the top kernels of a rocprofv3 --stats csv: kstats.py <csv> [n] [filter]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
flt = sys.argv[3] if len(sys.argv) > 3 else ""
tot = sum(float(r['TotalDurationNs']) for r in rows)
steps = 17.0
print(f"total kernel time {tot/1e6:.2f} ms = {tot/1e6/steps:.3f} ms/step over {steps:.0f} steps")
for r in [r for r in rows if flt in r['Name']][:n]:
    print(f"{r['Name'][:84]:84s} {int(r['Calls'])/steps:6.1f}/step avg {float(r['AverageNs'])/1e3:7.1f} us {float(r['TotalDurationNs'])/1e3/steps:8.1f} us/step {float(r['Percentage']):5.2f}%")
