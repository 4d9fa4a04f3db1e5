#!/bin/bash
# round 6, call 1: baseline, Legendre row-group sweep on one rank, shard_compute under measurement variants, rank 0 of the 8-rank job under rocprofv3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06a; mkdir -p $OUT
bash tools/ab_env.sh r06a/ab "T85L40" 2 - ISCA_LEG_NTG=1 ISCA_LEG_NTG=2 2>&1 | tee $OUT/ab.log
timeout 900 python tools/shard_ab.py $OUT T85L40 "4 8" nokeep,GPU_MAX_HW_QUEUES=4 nokeep keep,GPU_MAX_HW_QUEUES=4 keep keep,ISCA_LEG_NTG=3 keep,prof 2>&1 | tee $OUT/shard_T85.log
timeout 600 python tools/shard_ab.py $OUT T170L60 "8" nokeep,GPU_MAX_HW_QUEUES=4 keep keep,ISCA_LEG_NTG=3 2>&1 | tee $OUT/shard_T170.log
find $OUT -name "*kernel_trace.csv" -size +20M -delete
ls -R $OUT | head -50
