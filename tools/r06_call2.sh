#!/bin/bash
# round 6, call 2: the GPU suite with the pure-sigma column kernel, then A/B of the column kernel variants at T85L40 and T170L60
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06b; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest.log
bash tools/ab_env.sh r06b/ab "T85L40" 2 - ISCA_COLUMN_TWO=1 ISCA_COLUMN_GENERIC=1 2>&1 | tee $OUT/ab.log
bash tools/ab_env.sh r06b/ab170 "T170L60" 1 - ISCA_COLUMN_GENERIC=1 2>&1 | tee $OUT/ab170.log
