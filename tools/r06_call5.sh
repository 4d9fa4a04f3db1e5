#!/bin/bash
# round 6, call 5: suite on the product build; upper bound of retiring k_fixer_finish (experiments build); kernel trace of the headline
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06e; mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -q --maxfail=6 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest.log
export ISCA_DYN_LIB=$GRAFT_REPO_ROOT/isca_amd/lib/libisca_dyn_exp.so
bash tools/ab_env.sh r06e/ab "T85L40" 3 - ISCA_X_NO_FINISH=1 2>&1 | tee $OUT/ab.log
unset ISCA_DYN_LIB
export ISCA_BENCH_NO_EXTRA=1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python bench.py --workload T85L40 --steps 500 --warmup 50 --cpu-steps 0 > $OUT/bench_stats.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r06e/stats/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r['Name'].split('(')[0][-50:].ljust(52), r['Calls'], round(float(r['AverageNs'])/1e3,2))
PY
find $OUT -name "*kernel_trace.csv" -delete
