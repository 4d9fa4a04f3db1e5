#!/bin/bash
# moist kernel A/B inside one gpurun call: isca_amd/lib/libisca_dyn_old.so (a build of the commit before) against the current library, twice each
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for lib in old new; do
    if [ $lib = old ]; then export ISCA_DYN_LIB=$PWD/isca_amd/lib/libisca_dyn_old.so; else unset ISCA_DYN_LIB; fi
    echo "== $lib"; timeout 400 python tools/dev/moist_ab.py 2>&1 | tail -2 | cut -c1-330
  done
done
unset ISCA_DYN_LIB
[ -n "$1" ] && { timeout 1500 python -m pytest tests/test_gpu_moist.py -m gpu -x -q 2>&1 | tail -3; }
