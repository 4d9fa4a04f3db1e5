#!/bin/bash
# round 6, final collection: the GPU suite, profiles (tools/collect_profiles.sh), the parity table and the parity tests' printed figures
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06k
timeout 2400 python -m pytest tests -m gpu -q --maxfail=8 > gpurun_out/r06k/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r06k/pytest.log
bash tools/collect_profiles.sh
TOP=gpurun_out/prof_final
timeout 600 python tools/parity_report.py > $TOP/parity_table.md 2> $TOP/parity_table.err
timeout 900 python -m pytest tests -m gpu -q -s -k "trip or one_day or ten_days or golden_T85L40 or golden_T170L60 or developed or moist_trajectory" > $TOP/parity_prints.log 2>&1
tail -3 $TOP/parity_prints.log
