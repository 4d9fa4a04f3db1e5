#!/bin/bash
# Collect the rocprofv3 evidence for profiles/ on the GPU box:
#   1. kernel trace + stats of the headline bench command
#   2. separate --pmc passes (FETCH_SIZE, WRITE_SIZE; SQ wave/wait/VALU/MFMA counters) with --kernel-trace only
# Outputs land in gpurun_out/prof_final/ ; tools/summarize_profiles.py turns them into profiles/r01_*.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export ISCA_BENCH_NO_EXTRA=1     # only the headline workload in the profiled command
OUT=gpurun_out/prof_final
mkdir -p $OUT
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python bench.py --steps 500 --warmup 50 --cpu-steps 0 > $OUT/bench_stats.log 2>&1
i=0
for pm in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $pm --output-format csv -d $OUT/pmc_$i -o p -- python bench.py --steps 50 --warmup 10 --cpu-steps 0 > $OUT/pmc_$i.log 2>&1
done
python tools/summarize_profiles.py $OUT
