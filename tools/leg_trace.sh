#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export ISCA_BENCH_NO_EXTRA=1 ISCA_DYN_LIB=$GRAFT_REPO_ROOT/isca_amd/lib/libisca_dyn_trace.so ISCA_LEG_TRACE_AT=${2:-60}
mkdir -p gpurun_out
for W in ${1:-T170L60 T85L40}; do
  timeout 120 python bench.py --workload $W --steps 40 --warmup 20 --cpu-steps 0 > gpurun_out/trace_$W.log 2>&1
  tail -c 300 gpurun_out/trace_$W.log
  for d in fwd inv; do mv gpurun_out/leg_trace_$d.bin gpurun_out/leg_trace_${d}_$W.bin; echo "== $W $d"; python tools/dev/leg_trace.py gpurun_out/leg_trace_${d}_$W.bin; done
done
