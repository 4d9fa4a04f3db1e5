#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06n; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q --maxfail=8 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest.log
timeout 600 python -m pytest tests -m gpu -q -s -k "golden_T85L40 or golden_T170L60" 2>&1 | grep "vs the reference" | cut -c1-400
