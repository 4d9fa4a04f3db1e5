#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06i; mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -q --maxfail=8 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest.log
export ISCA_DYN_LIB=$GRAFT_REPO_ROOT/isca_amd/lib/libisca_dyn_exp.so
bash tools/ab_env.sh r06i/ab "T85L40" 3 - ISCA_COLUMN_TWO=0 ISCA_NO_DEFERRED_FINISH=1 2>&1 | tee $OUT/ab.log
bash tools/ab_env.sh r06i/ab170 "T170L60" 2 - ISCA_NO_DEFERRED_FINISH=1 2>&1 | tee $OUT/ab170.log
