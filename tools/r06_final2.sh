#!/bin/bash
# round 6, after the moist kernel's change: the GPU suite, the Frierson configuration's profile files (collect_profiles.sh ONLY=moist), the plain bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06p
timeout 2400 python -m pytest tests -m gpu -q --maxfail=8 > gpurun_out/r06p/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r06p/pytest.log
ONLY=moist bash tools/collect_profiles.sh
TOP=gpurun_out/prof_final
timeout 600 python bench.py --steps 500 --warmup 50 > $TOP/bench_T85L40.json.log 2>&1
tail -c 600 $TOP/bench_T85L40.json.log
timeout 600 python tools/parity_report.py > $TOP/parity_table.md 2> $TOP/parity_table.err
timeout 900 python -m pytest tests -m gpu -q -s -k "trip or one_day or ten_days or golden_T85L40 or golden_T170L60 or developed or moist_trajectory or sigma_log" > $TOP/parity_prints.log 2>&1
tail -3 $TOP/parity_prints.log
