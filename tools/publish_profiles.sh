#!/bin/bash
# Copy the summaries produced by tools/collect_profiles.sh (gpurun_out/prof_final) into profiles/ under this round's names.
#   usage: bash tools/publish_profiles.sh [round-tag, default r03]
set -e
cd "$(dirname "$0")/.."
R=${1:-r06}
T=gpurun_out/prof_final
for W in T85L40 T170L60 T85L40_moist T85L40_deferred_finish; do
  S=$T/$W
  [ -d $S ] || continue
  cp $S/kernel_stats_summary.csv profiles/${R}_${W}_kernel_stats.csv
  cp $S/stats/bench_kernel_stats.csv profiles/${R}_${W}_kernel_stats_rocprofv3_raw.csv
  if [ -f $S/pmc_summary.csv ] && [ $(wc -l < $S/pmc_summary.csv) -gt 1 ]; then
    cp $S/pmc_summary.csv profiles/${R}_${W}_pmc_summary.csv
    if [ $W = T85L40 ]; then cp $S/pmc_traffic.json profiles/${R}_pmc_traffic.json; else cp $S/pmc_traffic.json profiles/${R}_${W}_pmc_traffic.json; fi
  fi
  if grep -q '^{' $S/bench_stats.log; then grep '^{' $S/bench_stats.log | tail -1 > profiles/${R}_${W}_bench_under_rocprof.json; fi      # the moist run prints no bench line
done
grep '^{' $T/bench_T85L40.json.log | tail -1 > profiles/${R}_T85L40_bench.json
# the sharded step's compute (tools/shard_ab.py): the HIP-event figures, and rank 0's rocprofv3 kernel statistics per rank count
python - $T/shard $R <<'PY'
import csv, glob, json, os, sys
src, R = sys.argv[1], sys.argv[2]
out = {}
for f in sorted(glob.glob(os.path.join(src, "shard_ab_*.json"))):
    wl = os.path.basename(f)[len("shard_ab_"):-5]
    d = json.load(open(f))
    out[wl] = d.get("-,prof") or next(iter(d.values()))
if out:
    json.dump(out, open(f"profiles/{R}_shard_compute.json", "w"), indent=1)
for d in sorted(glob.glob(os.path.join(src, "rocprof_*", "*_P*"))):
    tag = os.path.basename(d)                       # T85L40_P8
    f = glob.glob(os.path.join(d, "*kernel_stats.csv"))
    if not f:
        continue
    with open(f"profiles/{R}_{tag}_kernel_stats.csv", "w") as o:
        o.write("kernel,calls,avg_us,min_us,max_us\n")
        for r in csv.DictReader(open(f[0])):
            n = r["Name"].replace("(anonymous namespace)::", "")
            if int(r["Calls"]) < 20 or n.startswith("__amd"):
                continue
            o.write(f"\"{n.split('(')[0].replace('void ', '')}\",{r['Calls']},{float(r['AverageNs']) / 1e3:.2f},{float(r['MinNs']) / 1e3:.2f},{float(r['MaxNs']) / 1e3:.2f}\n")
PY
# the derived table is ALWAYS regenerated from the files just copied (a table older than its sources is not evidence)
python tools/roofline_table.py $R > profiles/${R}_roofline_table.md
ls -la profiles/ | grep ${R}_
