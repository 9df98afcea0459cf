"""profiles/ artefacts from a scratch/profile_round.sh run: python scratch/make_profile_docs.py prof_v10 r01_v10"""
import csv, json, os, re, sys, shutil
src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
S = os.path.join(ROOT, "gpurun_out", src)
P = os.path.join(ROOT, "profiles")
shutil.copy(os.path.join(S, "trace", "t_kernel_stats.csv"), os.path.join(P, f"{tag}_bench_kernel_stats.csv"))
shutil.copy(os.path.join(S, "pmc_summary.csv"), os.path.join(P, f"{tag}_pmc_hbm_summary.csv"))
line = [l for l in open(os.path.join(S, "trace.log")) if l.startswith('{"metric"')]
bench = json.loads(line[-1]) if line else {}
rows = list(csv.DictReader(open(os.path.join(S, "trace", "t_kernel_stats.csv"))))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
steps = bench.get("steps", 10) + bench.get("warmup", 2) + 2 + 3   # timed + warm-up + instrumented eager + graph warm-up
pm = list(csv.DictReader(open(os.path.join(S, "pmc_summary.csv"))))
def pmc(counter, pat):
    n = t = 0
    for r in pm:
        if r["counter"] == counter and re.search(pat, r["kernel"]):
            n += int(r["launches"]); t += int(r["launches"]) * float(r["mean_value_KB"])
    return n, (t / n if n else 0.0)
nt = r"gemm_bf16_dma_kernel<false, false"
nf, f = pmc("FETCH_SIZE", nt); nw, w = pmc("WRITE_SIZE", nt)
traffic = {"kernel": "gemm_bf16_dma_kernel<false,false,64,2,{4|5}> (NT), all launches of one train step (mixed epilogues)",
           "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (two separate passes) over `python bench.py --steps 2 --warmup 1 "
                     "--no-graph --no-cpu-baseline`; launch-weighted mean",
           "launches": nf, "FETCH_SIZE_KB": round(f, 2), "WRITE_SIZE_KB": round(w, 2),
           "correction": "gfx950: FETCH_SIZE counts 128-B requests at 64 B for wide coalesced reads -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact",
           "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
m2 = r"gemm_bf16_dma2_kernel"
nf2, f2 = pmc("FETCH_SIZE", m2); nw2, w2 = pmc("WRITE_SIZE", m2)
if nf2:
    traffic["merged"] = {"kernel": "gemm_bf16_dma2_kernel<{4|5}> (dW slabs + dX + previous layer's slab reduction), all launches of one train step",
                         "launches": nf2, "FETCH_SIZE_KB": round(f2, 2), "WRITE_SIZE_KB": round(w2, 2),
                         "hbm_bytes_per_launch": int((2 * f2 + w2) * 1024)}
json.dump(traffic, open(os.path.join(P, "gemm_traffic.json"), "w"), indent=1)
with open(os.path.join(P, f"{tag}_summary.md"), "w") as o:
    o.write(f"# Round 1, {tag} -- rocprofv3 --kernel-trace --stats of the default bench (hipGraph replay)\n\n")
    o.write("Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline` "
            "(1x MI355X, cfg2, bf16, dropout 0.2); PMC: two more passes with `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` over the eager "
            "(`--no-graph`) run, see `scratch/profile_round.sh`.\n\n")
    if bench:
        o.write(f"Bench line of the profiled run: {bench['ms_per_step']} ms/step, {bench['value']} clips/s; roofline "
                f"{json.dumps(bench.get('roofline'))}\n\n")
    o.write(f"Kernel time summed over the run: {tot/1e6:.1f} ms.\n\n| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
    for r in rows[:32]:
        o.write(f"| `{r['Name'][:100]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e3:.1f} | {float(r['Percentage']):.1f} |\n")
    o.write(f"\nHBM traffic of the NT LDS-DMA GEMM (launch-weighted mean over {nf} launches): FETCH_SIZE {f:.0f} KB x2 (gfx950 correction) "
            f"+ WRITE_SIZE {w:.0f} KB = {traffic['hbm_bytes_per_launch']/1e6:.1f} MB per launch (`gemm_traffic.json`).\n")
print(json.dumps(traffic)); print("bench:", bench.get("ms_per_step"), bench.get("roofline"))
