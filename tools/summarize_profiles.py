#!/usr/bin/env python3
"""Summarise gpurun_out/prof_final into small, commit-able files (run on the GPU box after collect_profiles.sh)."""
import csv, glob, json, os, sys, collections
src = sys.argv[1]
out = {}
stats = glob.glob(os.path.join(src, "stats", "*kernel_stats.csv"))
rows = []
if stats:
    for r in csv.DictReader(open(stats[0])):
        rows.append({"kernel": r["Name"], "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                     "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3, "pct": float(r["Percentage"])})
tot = collections.defaultdict(dict)
for d in sorted(glob.glob(os.path.join(src, "pmc_*/"))):
    f = glob.glob(d + "*counter_collection.csv")
    if not f:
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f[0])):
        acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in acc.items():
        for c, v in cs.items():
            tot[k][c] = sum(v) / len(v)
def short(name):
    n = name.replace("(anonymous namespace)::", "").split("(")[0]
    n = n.replace("void ", "").replace("isca::", "")
    base = n.split("<")[0]
    if base == "k_leg_inv_coop" and "<" in n:       # <NW, JTG, FUSED, WPS>: the step's fused synthesis vs the staged one of the API transforms
        args = [a.strip() for a in n.split("<", 1)[1].rstrip(">").split(",")]
        base += ":fused" if args[2] == "true" else ":staged"
    return base
bytes_per_launch = {}
for k, cs in tot.items():
    if "isca::" in k and "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        # rocprofv3 reports KiB; gfx950: FETCH_SIZE counts 128-B requests as 64 B for coalesced streaming reads -> x2
        # (calibrated here on k_column: 10 L-level fields + 4 2-D read = 106 MB known, FETCH_SIZE = 52.8 MB)
        bytes_per_launch[short(k)] = (2.0 * cs["FETCH_SIZE"] + cs["WRITE_SIZE"]) * 1024.0
json.dump({"bytes_per_launch": bytes_per_launch, "note": "2*FETCH_SIZE + WRITE_SIZE (KiB) per launch, averaged over the run"},
          open(os.path.join(src, "pmc_traffic.json"), "w"), indent=1)
with open(os.path.join(src, "kernel_stats_summary.csv"), "w") as f:
    f.write("kernel,calls,avg_us,min_us,max_us,pct\n")
    for r in rows:
        if r["calls"] >= 50:
            f.write(f"\"{short(r['kernel'])}\",{r['calls']},{r['avg_us']:.2f},{r['min_us']:.2f},{r['max_us']:.2f},{r['pct']:.2f}\n")
with open(os.path.join(src, "pmc_summary.csv"), "w") as f:
    keys = sorted({c for cs in tot.values() for c in cs})
    f.write("kernel," + ",".join(keys) + "\n")
    for k, cs in tot.items():
        if "isca::" in k and cs.get("SQ_WAVES", 0) >= 1:
            f.write(f"\"{short(k)}\"," + ",".join(f"{cs.get(c, float('nan')):.0f}" for c in keys) + "\n")
print(open(os.path.join(src, "kernel_stats_summary.csv")).read())
print(json.dumps(bytes_per_launch, indent=1))
