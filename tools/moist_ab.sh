#!/bin/bash
# One GPU-box visit: the moist tests, then tools/dev/moist_ab.py once per variant.  usage (through gpurun): bash tools/moist_ab.sh <tag> <pytest -k expr or ""> variant [variant ...]   ("-" = defaults; NAME=VALUE,NAME=VALUE)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; KEXPR=$2; shift; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
if [ -n "$KEXPR" ]; then
  timeout 900 python -m pytest tests/test_gpu_moist.py -m gpu -x -q -k "$KEXPR" > $OUT/pytest.log 2>&1
  echo "pytest rc=$?"; tail -4 $OUT/pytest.log
fi
for V in "$@"; do
  ( [ "$V" != "-" ] && export ${V//,/ }
    timeout 300 python tools/dev/moist_ab.py 2>&1 | tail -3 | tee -a $OUT/ab.log )
done
