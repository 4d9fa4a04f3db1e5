#!/bin/bash
# round 6, call 4: suite; T170L60 column kernel with two blocks per CU; sharded compute after the reverted fixer fold
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06d; mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -q --maxfail=6 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest.log
bash tools/ab_env.sh r06d/ab "T85L40" 2 - ISCA_COLUMN_TWO=0 2>&1 | tee $OUT/ab.log
bash tools/ab_env.sh r06d/ab170 "T170L60" 2 - ISCA_COLUMN_TWO=1 2>&1 | tee $OUT/ab170.log
timeout 900 python tools/shard_ab.py $OUT T85L40 "2 4 8" - 2>&1 | tee $OUT/shard_T85.log
timeout 600 python tools/shard_ab.py $OUT T170L60 "4 8" - 2>&1 | tee $OUT/shard_T170.log
