mkdir -p gpurun_out/r4; python bench.py --legs exact,chw,336 2>/dev/null | tail -1 > gpurun_out/r4/bench_lease.json; python - <<PY
import json
d=json.load(open("gpurun_out/r4/bench_lease.json"))
r=d["roofline"]; print("main", d["value"], d["ms_per_step"], r["frac"], "add", r["calibration"]["frac"])
for k in d:
    if k.startswith("also"):
        v=d[k]; print(k, v.get("ms_per_step"), (v.get("roofline") or {}).get("frac"), v.get("step_frac_of_hbm_peak"))
PY
