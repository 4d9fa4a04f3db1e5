#!/bin/bash
# headline workload A/B inside one gpurun call: isca_amd/lib/libisca_dyn_old.so (a build of the commit before) against the current library, NREP times each;
# extra arguments: NAME=VALUE variants of the current library
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export ISCA_BENCH_NO_EXTRA=1
W=${WORKLOAD:-T85L40}; S=${STEPS:-2000}
run() { python bench.py --workload $W --steps $S --warmup 100 --cpu-steps 0 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('  %.4f ms/step' % d['ms_per_step'], ' '.join('%s %.1f' % (k, v * 1e3) for k, v in d['kernel_ms'].items()))"; }
for rep in $(seq ${NREP:-2}); do
  echo "== old"; ( export ISCA_DYN_LIB=$PWD/isca_amd/lib/libisca_dyn_old.so; run )
  echo "== new"; run
  for V in "$@"; do echo "== new $V"; ( export ${V//,/ }; run ); done
done
