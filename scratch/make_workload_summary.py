"""profiles/<tag>_summary.md + <tag>_kernel_stats.csv from one `scratch/profile_workload.sh <workload> <dir>` run:
python scratch/make_workload_summary.py gpurun_out/<dir> <tag> "<title>".  Steps traced = calls of the AdamW kernel."""
import csv, json, os, re, shutil, sys

src, tag, title = sys.argv[1], sys.argv[2], sys.argv[3]
stats = os.path.join(src, "trace", "t_kernel_stats.csv")
rows = list(csv.DictReader(open(stats)))
line = [l for l in open(os.path.join(src, "trace.log")) if l.startswith('{"metric"')][-1]
bench = json.loads(line)
steps = next(int(r["Calls"]) for r in rows if "adamw_kernel" in r["Name"])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
calls = sum(int(r["Calls"]) for r in rows)
glue = sum(float(r["TotalDurationNs"]) for r in rows if re.search(r"at::|rocclr|Cijk|elementwise", r["Name"]) and "Cijk" not in r["Name"])
out = [f"# {tag} -- {title}", "",
       "Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --workload "
       f"{bench['config']['workload'].split(' ')[0].lower()} --steps 50 --warmup 2 --no-cpu-baseline` (scratch/profile_workload.sh, scratch/make_workload_summary.py).", "",
       f"Bench line of the traced run: {bench['ms_per_step']} ms per step, {bench['value']} {bench['unit']}.  Kernel time {tot / steps / 1e6:.2f} ms per step over "
       f"{steps} steps, {calls / steps:.0f} launches per step; torch glue (at::* + copies) {100 * glue / tot:.1f} % = {glue / steps / 1e6:.2f} ms per step.", "",
       "| kernel | calls/step | avg us | ms/step | % |", "|---|---|---|---|---|"]
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:45]:
    t = float(r["TotalDurationNs"])
    out.append(f"| `{r['Name'][:100]}` | {int(r['Calls']) / steps:.1f} | {float(r['AverageNs']) / 1e3:.1f} | {t / steps / 1e6:.2f} | {100 * t / tot:.1f} |")
open(f"profiles/{tag}_summary.md", "w").write("\n".join(out) + "\n")
shutil.copy(stats, f"profiles/{tag}_kernel_stats.csv")
print("\n".join(out[:24]))
