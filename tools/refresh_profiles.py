#!/usr/bin/env python3
"""Copy the results of tools/pmc_bench.sh TAG (gpurun_out/TAG_bench) into profiles/ and print the numbers the
documents quote: tools/refresh_profiles.py [TAG]"""
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = os.path.join("gpurun_out", f"{tag}_bench")
for name in (f"{tag}_usm_pmc.json", f"{tag}_usm_pmc_sq.csv", f"{tag}_usm_pmc_fetch.csv", f"{tag}_usm_pmc_write.csv",
             f"{tag}_usm_pmc_mix.csv"):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join("profiles", name))
shutil.copy(os.path.join(src, "bench.json"), os.path.join("profiles", f"{tag}_bench_line.json"))
shutil.copy(os.path.join(src, "bench_traced.json"), os.path.join("profiles", f"{tag}_bench_line_under_rocprof.json"))
def stats(trace_dir, out_name):
    """Kernel statistics of a traced run, the framework's kilobyte-long kernel names cut short."""
    rows = list(csv.DictReader(open(os.path.join(src, trace_dir, "t_kernel_stats.csv"))))
    with open(os.path.join("profiles", out_name), "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(rows[0].keys())
        for r in rows:
            if len(r["Name"]) > 160:
                r["Name"] = r["Name"][:157] + "..."
            w.writerow(r.values())
    return rows


# the headline workload alone (every usm_kernel launch is the headline launch) and the full default run beside it
rows = stats("trace_headline", f"{tag}_bench_kernel_stats.csv")
stats("trace", f"{tag}_bench_kernel_stats_full_run.csv")
shutil.copy(os.path.join(src, "bench_traced_headline.json"), os.path.join("profiles", f"{tag}_bench_line_headline_under_rocprof.json"))
u = [r for r in rows if "usm_kernel" in r["Name"]][0]
d = json.load(open(os.path.join("profiles", f"{tag}_bench_line.json")))
pm = json.load(open(os.path.join("profiles", f"{tag}_usm_pmc.json")))
print("value", d["value"], "ms/step", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"],
      "achieved", d["roofline"]["achieved"], "valu_frac", d["roofline"].get("valu_frac"), "valu_stall_frac",
      d["roofline"].get("valu_stall_frac"))
print("trace: usm_kernel", u["Calls"], "launches, average", float(u["AverageNs"]) / 1e6, "ms")
print("pmc:", {k: pm[k] for k in ("clock_ghz", "valu_per_px", "trans_per_px", "valu_busy", "fetch_bytes_per_px", "write_bytes_per_px")})
for e in d["others"]:
    print(f'{e["kernel_ms"]:9.4f} ms  frac {e["frac"]:.4f}  {e["config"][:80]}')
print("cpu:", d["cpu_baseline"]["value"], d["cpu_baseline"]["threads4"]["value"], d["cpu_baseline"]["threads_all"]["value"])
