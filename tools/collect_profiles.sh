#!/bin/bash
# Collect the rocprofv3 evidence for profiles/ on the GPU box (usage: bash tools/collect_profiles.sh [round-tag]):
#   1. kernel trace + stats of the headline bench command (T85L40), of the T170L60 workload and of the moist (Frierson) configuration
#   2. separate --pmc passes (FETCH_SIZE, WRITE_SIZE; SQ wave/wait/VALU/MFMA counters) with --kernel-trace only, T85L40 and T170L60
# Outputs land in gpurun_out/prof_final/<workload>/ ; tools/summarize_profiles.py turns them into small files, tools/publish_profiles.sh
# copies those into profiles/ under the round's names.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export ISCA_BENCH_NO_EXTRA=1     # only the named workload in the profiled command
TOP=gpurun_out/prof_final
rm -rf $TOP; mkdir -p $TOP $TOP/shard
WL="T85L40 T170L60"; [ "$ONLY" = moist ] && WL=""        # ONLY=moist: the Frierson configuration's files alone
for W in $WL; do
  OUT=$TOP/$W; mkdir -p $OUT
  S=500; [ $W = T170L60 ] && S=150
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python bench.py --workload $W --steps $S --warmup 50 --cpu-steps 0 > $OUT/bench_stats.log 2>&1
  i=0
  for pm in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 200 rocprofv3 --kernel-trace --pmc $pm --output-format csv -d $OUT/pmc_$i -o p -- python bench.py --workload $W --steps 40 --warmup 10 --cpu-steps 0 > $OUT/pmc_$i.log 2>&1
  done
  python tools/summarize_profiles.py $OUT
  find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete      # the raw traces: gpurun brings back 64 MiB at most
done
# the headline once more with the fixers' finish deferred into block 0 of the next column kernel (experiments build: ISCA_DEFERRED_FINISH=1; HISTORY "Round 6")
if [ -f isca_amd/lib/libisca_dyn_exp.so ] && [ "$ONLY" != moist ]; then
  OUT=$TOP/T85L40_deferred_finish; mkdir -p $OUT
  ISCA_DYN_LIB=$GRAFT_REPO_ROOT/isca_amd/lib/libisca_dyn_exp.so ISCA_DEFERRED_FINISH=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python bench.py --workload T85L40 --steps 500 --warmup 50 --cpu-steps 0 > $OUT/bench_stats.log 2>&1
  python tools/summarize_profiles.py $OUT > /dev/null
  find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
fi
# moist configuration (BASELINE configs[3] at T85L40) in a spun-up state: the profiler collects for 2 s from second 9 of a run that spins up
# 10 000 steps (35 days: it rains; the moist kernel is 25 % slower than in the first days after the cold start) and then keeps stepping
OUT=$TOP/T85L40_moist; mkdir -p $OUT
timeout 300 rocprofv3 --collection-period 9:2:1 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python tools/dev/moist_bench.py T85 40 300 spunup > $OUT/bench_stats.log 2>&1
i=0
for pm in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD"; do      # HBM traffic and issue counters of the moist column kernel
  i=$((i+1))
  # (the counter passes are of steps 60-100 after the cold start: rocprofv3 --pmc with --collection-period dumped core, and 10 000 steps under --pmc take too long)
  timeout 200 rocprofv3 --kernel-trace --pmc $pm --output-format csv -d $OUT/pmc_$i -o p -- python tools/dev/moist_bench.py T85 40 300 short > $OUT/pmc_$i.log 2>&1
done
python tools/summarize_profiles.py $OUT
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete
[ "$ONLY" = moist ] && exit 0
# the sharded step's compute: P processes taking turns on this GPU (bench.py: shard_compute), rank 0 of each job under rocprofv3 --kernel-trace --stats
unset ISCA_BENCH_NO_EXTRA
timeout 900 python tools/shard_ab.py $TOP/shard T85L40 "2 4 8" -,prof > $TOP/shard/T85L40.log 2>&1
timeout 600 python tools/shard_ab.py $TOP/shard T170L60 "4 8" -,prof > $TOP/shard/T170L60.log 2>&1
find $TOP/shard -name "*kernel_trace.csv" -delete; find $TOP/shard -name "*agent_info.csv" -delete
# the plain bench line of the headline workload (no profiler attached)
unset ISCA_BENCH_NO_EXTRA
timeout 600 python bench.py --steps 500 --warmup 50 > $TOP/bench_T85L40.json.log 2>&1
tail -c 600 $TOP/bench_T85L40.json.log
