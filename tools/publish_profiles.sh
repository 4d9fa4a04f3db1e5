#!/bin/bash
# Copy the summaries produced by tools/collect_profiles.sh (gpurun_out/prof_final) into profiles/ under this round's names.
#   usage: bash tools/publish_profiles.sh [round-tag, default r01]
set -e
cd "$(dirname "$0")/.."
R=${1:-r01}
S=gpurun_out/prof_final
cp $S/kernel_stats_summary.csv profiles/${R}_T85L40_kernel_stats.csv
cp $S/stats/bench_kernel_stats.csv profiles/${R}_T85L40_kernel_stats_rocprofv3_raw.csv
cp $S/pmc_summary.csv profiles/${R}_T85L40_pmc_summary.csv
cp $S/pmc_traffic.json profiles/${R}_pmc_traffic.json
grep '^{' $S/bench_stats.log | tail -1 > profiles/${R}_T85L40_bench_under_rocprof.json
ls -la profiles/
