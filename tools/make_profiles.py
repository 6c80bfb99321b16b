"""Turn gpurun_out/<tag>/ (written by tools/refresh_profiles.sh on the GPU box) into the committed summaries
under profiles/: kernel-stats table, HBM traffic of the resample kernel from the PMC passes (both arithmetic modes),
bench lines, stage / chain benches.   usage: make_profiles.py [src_tag] [name]   (default r2 round2)"""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pos = [a for a in sys.argv[1:] if not a.startswith("--")]
src_tag = pos[0] if len(pos) > 0 else "r6"
tag = pos[1] if len(pos) > 1 else "round6"
SRC = os.path.join(ROOT, "gpurun_out", src_tag)
DST = os.path.join(ROOT, "profiles")

def one(pattern):
    files = glob.glob(os.path.join(SRC, pattern), recursive=True)
    assert files, pattern
    return max(files, key=os.path.getmtime)      # gpurun merges into gpurun_out/: older leases' files may still lie there

def pmc(dirname, counter):
    vals = []
    for r in csv.DictReader(open(one(f"keep/{dirname}/*counter_collection.csv"))):
        if "remap_rows_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter:
            vals.append(float(r["Counter_Value"]))
    return vals
B, S = 256, 1024
alg = 2 * B * S * S * 3 * 4
traffic = {}
for mode in ("cv2", "exact", "cv2_chw"):
    fetch, write = pmc(f"pmc_fetch_{mode}", "FETCH_SIZE"), pmc(f"pmc_write_{mode}", "WRITE_SIZE")
    f_kb, w_kb = sum(fetch) / len(fetch), sum(write) / len(write)
    total = (2 * f_kb + w_kb) * 1024
    traffic[f"1024_{mode}"] = {
        "remap_rows_kernel_bytes_per_launch": total, "ratio_to_algorithmic": total / alg, "FETCH_SIZE_KB_raw": f_kb,
        "WRITE_SIZE_KB_raw": w_kb, "launches_averaged": min(len(fetch), len(write)),
        "note": f"rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python bench.py --no-cpu-baseline "
                f"--no-also --steps 5 --mode {mode.split('_')[0]}{' --layout chw' if mode.endswith('_chw') else ''}`, mean over the launches of remap_rows_kernel; FETCH_SIZE doubled per "
                f"MI355X_MICROARCH.md (gfx950 reports 1/2 of a wide coalesced streaming read); KB -> bytes x1024"}
    print(mode, "traffic ratio", total / alg)
lease = open(os.path.join(SRC, "lease.txt")).read().strip().replace("\n", "; ") if os.path.exists(os.path.join(SRC, "lease.txt")) else "unrecorded"
for v in traffic.values():
    v["lease"] = lease
    v["same_lease_as"] = (f"profiles/{tag}_bench.json, {tag}_bench_kernel_stats.* (one run of tools/refresh_profiles.sh {src_tag}: the counter passes "
                          "ran FIRST, the bench line that quotes them right behind on the same box)")
json.dump(traffic, open(os.path.join(DST, "pmc_traffic.json"), "w"), indent=1)
if "--pmc-only" in sys.argv:
    # (refresh_profiles.sh runs this ON THE GPU BOX between the counter passes and the bench line, so that the line's
    #  `roofline.traffic` -- read from profiles/pmc_traffic.json -- quotes the lease it was measured on)
    shutil.copy(os.path.join(DST, "pmc_traffic.json"), os.path.join(SRC, "pmc_traffic.json"))
    sys.exit(0)
stats = one("keep/trace/*kernel_stats.csv")
shutil.copy(stats, os.path.join(DST, f"{tag}_bench_kernel_stats.csv"))
full = glob.glob(os.path.join(SRC, "keep/trace_full/*kernel_stats.csv"))
if full:
    shutil.copy(max(full, key=os.path.getmtime), os.path.join(DST, f"{tag}_bench_full_kernel_stats.csv"))
rows = list(csv.DictReader(open(stats)))
prof = json.loads([l for l in open(os.path.join(SRC, "bench_profiled.json")) if l.startswith("{")][-1])
with open(os.path.join(DST, f"{tag}_bench_kernel_stats.md"), "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline ({tag})\n\n")
    f.write("Command (on the MI355X box): `rocprofv3 --kernel-trace --stats --output-format csv -d ... -- python3 bench.py "
            "--no-cpu-baseline --no-also` (tools/refresh_profiles.sh): ONLY the main workload, mode=cv2 HWC at B=256 1024x1024 -- "
            "the three-launch step (`also_eager`: 3 warm-up + 20 timed), then the MAIN line, the "
            "two-launch stream step (`attn_maps_kernel` + `remap_rows_kernel`: 2 priming launches, 5 + 3 warm-up + 20 timed "
            "steps) -- so the LAST 20 launches of `remap_rows_kernel<..., 1, true>` in the kernel trace are the ones the main "
            "line times (per-dispatch durations below the table); the table's average runs over all of them.  The stats of the "
            "WHOLE default command (exact / CHW / fused / peaked / zero attention / 336 workloads, whose launches share kernel "
            "names) are in `" + tag + "_bench_full_kernel_stats.csv`.\n\n")
    f.write("| kernel | calls | avg us | min us | max us | total ms | % |\n|---|---:|---:|---:|---:|---:|---:|\n")
    for r in rows:
        f.write(f"| `{r['Name'][:140]}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} | "
                f"{float(r['MaxNs'])/1e3:.1f} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['Percentage']):.2f} |\n")
    tr = glob.glob(os.path.join(SRC, "keep/trace/*kernel_trace.csv"))
    if tr:
        d = [(int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
             for r in csv.DictReader(open(max(tr, key=os.path.getmtime))) if "remap_rows_kernel" in r["Kernel_Name"]]
        d = [x for _, x in sorted(d)]
        last = d[-prof["steps"]:]
        alg = prof["roofline"]["algorithmic_bytes_per_launch"]
        shutil.copy(max(tr, key=os.path.getmtime), os.path.join(DST, f"{tag}_bench_kernel_trace.csv"))
        f.write(f"\nPer-dispatch durations of `remap_rows_kernel` from the kernel trace of the same run (`{tag}_bench_kernel_trace.csv`): all "
                f"{len(d)} launches average {sum(d) / len(d):.1f} us; the LAST {len(last)} launches -- the ones `bench.py` times -- average "
                f"**{sum(last) / len(last):.1f} us** (min {min(last):.1f}, max {max(last):.1f}) = {alg / (sum(last) / len(last) * 1e-6) / 8e12:.4f} of 8 TB/s; "
                f"the first {len(d) - len(last)} are the untimed warm-up / priming steps and the three-launch (`also_eager`) measurement.\n")
    f.write(f"\nbench.py's own HIP-event measurement in the same (profiled) process: ms_per_step {prof['ms_per_step']}, "
            f"roofline {json.dumps(prof['roofline'])}, "
            f"stages_ms {json.dumps(prof.get('stages_ms'))}.\n")

bench_line = [l for l in open(os.path.join(SRC, "bench.json")) if l.startswith("{")][-1]
open(os.path.join(DST, f"{tag}_bench.json"), "w").write(bench_line)
if os.path.exists(os.path.join(SRC, "pmc_traffic.json")):      # the file the bench line read on the box
    shutil.copy(os.path.join(SRC, "pmc_traffic.json"), os.path.join(DST, "pmc_traffic.json"))
for name in ("bench_336", "bench_336x256", "bench_main_batched", "bench_main_batched_ragged", "bench_config5", "bench_force_dist"):
    if os.path.exists(os.path.join(SRC, f"{name}.json")):
        open(os.path.join(DST, f"{tag}_{name}.json"), "w").write([l for l in open(os.path.join(SRC, f"{name}.json")) if l.startswith("{")][-1])
for name in ("stage_bench", "chain_bench", "probe_bench", "remap_bench", "chain_kernel_stats", "attn_bench", "u8_bench", "chain_stream",
             "pair_step", "chain_step_kernel_stats", "remap_lines", "timeline_chain_32_336_500", "timeline_chain_64_336_500",
             "timeline_chain_256_1024_500", "timeline_step_64_336", "timeline_step_256_336", "timeline_remap_256_1024",
             "timeline_remap_256_336", "timeline_ragged_32", "timeline_ragged_256", "bounds", "ragged_kernel_stats", "lease"):
    if os.path.exists(os.path.join(SRC, f"{name}.txt")):
        shutil.copy(os.path.join(SRC, f"{name}.txt"), os.path.join(DST, f"{tag}_{name}.txt"))
print(bench_line)
