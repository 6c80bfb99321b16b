"""Turn gpurun_out/r1/ (written by tools/refresh_profiles.sh on the GPU box) into the committed summaries
under profiles/: kernel-stats table, HBM traffic of the resample kernel from the PMC passes, bench lines."""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r1")
DST = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "round1"

def one(pattern):
    files = glob.glob(os.path.join(SRC, pattern), recursive=True)
    assert len(files) == 1, (pattern, files)
    return files[0]

stats = one("trace/**/*kernel_stats.csv")
shutil.copy(stats, os.path.join(DST, f"{tag}_bench_kernel_stats.csv"))
rows = list(csv.DictReader(open(stats)))
prof_line = [l for l in open(os.path.join(SRC, "bench_profiled.json")) if l.startswith("{")][-1]
prof = json.loads(prof_line)
with open(os.path.join(DST, f"{tag}_bench_kernel_stats.md"), "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline ({tag}, final code)\n\n")
    f.write("Command (on the MI355X box): `rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r1/trace -- "
            "python bench.py --no-cpu-baseline` (tools/refresh_profiles.sh)\n\n")
    f.write("| kernel | calls | avg us | min us | max us | total ms | % |\n|---|---:|---:|---:|---:|---:|---:|\n")
    for r in rows:
        f.write(f"| `{r['Name'][:120]}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} | "
                f"{float(r['MaxNs'])/1e3:.1f} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['Percentage']):.2f} |\n")
    f.write(f"\nbench.py's own HIP-event measurement in the same (profiled) process: ms_per_step {prof['ms_per_step']}, "
            f"roofline {json.dumps(prof['roofline'])}, stages_ms {json.dumps(prof.get('stages_ms'))}.\n")

def pmc(dirname, counter):
    vals = []
    for r in csv.DictReader(open(one(f"{dirname}/**/*counter_collection.csv"))):
        if "remap_rows_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter:
            vals.append(float(r["Counter_Value"]))
    return vals
fetch, write = pmc("pmc_fetch", "FETCH_SIZE"), pmc("pmc_write", "WRITE_SIZE")
B, S = 256, 1024
alg = 2 * B * S * S * 3 * 4
f_kb, w_kb = sum(fetch) / len(fetch), sum(write) / len(write)
total = (2 * f_kb + w_kb) * 1024
json.dump({"1024": {"remap_rows_kernel_bytes_per_launch": total, "ratio_to_algorithmic": total / alg,
                    "FETCH_SIZE_KB_raw": f_kb, "WRITE_SIZE_KB_raw": w_kb, "launches_averaged": min(len(fetch), len(write)),
                    "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python bench.py "
                            "--no-cpu-baseline --no-also --steps 5`, mean over the launches of remap_rows_kernel; FETCH_SIZE "
                            "doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of a wide coalesced streaming read); KB -> bytes x1024"}},
          open(os.path.join(DST, "pmc_traffic.json"), "w"), indent=1)
bench_line = [l for l in open(os.path.join(SRC, "bench.json")) if l.startswith("{")][-1]
open(os.path.join(DST, f"{tag}_bench.json"), "w").write(bench_line)
shutil.copy(os.path.join(SRC, "stage_bench.txt"), os.path.join(DST, f"{tag}_stage_bench.txt"))
shutil.copy(os.path.join(SRC, "probe_bench.txt"), os.path.join(DST, f"{tag}_probe_bench.txt"))
print(bench_line)
print("traffic ratio", total / alg)
