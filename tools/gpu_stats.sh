#!/bin/bash
# rocprofv3 kernel stats of a short headline bench run; prints the per-kernel averages.   usage (through gpurun): bash tools/gpu_stats.sh [workload] [env assignments...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
W=${1:-T85L40}; shift
for kv in "$@"; do export "$kv"; done
export ISCA_BENCH_NO_EXTRA=1
OUT=gpurun_out/stats_$W
rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python bench.py --workload $W --steps 300 --warmup 40 --cpu-steps 0 > $OUT/bench.log 2>&1
python - $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/stats/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "isca" in r["Name"]:
        print("%-60s calls %6s avg %8.2f us" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
grep '^{' $OUT/bench.log | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('ms/step', d['ms_per_step'])"
