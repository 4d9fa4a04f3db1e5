#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06h; mkdir -p $OUT
export ISCA_DYN_LIB=$GRAFT_REPO_ROOT/isca_amd/lib/libisca_dyn_exp.so
bash tools/ab_env.sh r06h/ab "T85L40" 3 - ISCA_COLUMN_TWO=0 ISCA_NO_DEFERRED_FINISH=1 ISCA_NO_DEFERRED_FINISH=1,ISCA_COLUMN_TWO=0 2>&1 | tee $OUT/ab.log
unset ISCA_DYN_LIB
timeout 900 python tools/shard_ab.py $OUT T85L40 "4 8" - 2>&1 | tee $OUT/shard_T85.log
timeout 900 python -m pytest tests -m gpu -q --maxfail=8 -k "sharded or shard" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
