#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06l; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_moist.py tests/test_gpu_fortran_dropin.py -m gpu -q --maxfail=8 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
timeout 600 python tools/dev/moist_bench.py T85 40 300 spunup 2>&1 | tail -5
