#!/bin/bash
# One GPU-box visit: the -m gpu tests, then short bench runs (kernel timings included) of the headline and the T170L60 workloads.
#   usage (through gpurun): bash tools/gpu_check.sh [tag] [pytest args...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-check}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export ISCA_BENCH_NO_EXTRA=1
timeout 900 python -m pytest tests -m gpu -x -q "$@" > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 $OUT/pytest.log
for W in T85L40 T170L60; do
  timeout 300 python bench.py --workload $W --steps 200 --warmup 30 --cpu-steps 0 > $OUT/bench_$W.log 2>&1
  python - $OUT/bench_$W.log <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    if ln.startswith("{"):
        d = json.loads(ln)
        print(d["config"]["workload"][:8], "ms/step", round(d["ms_per_step"], 4), "SYPD", round(d["value"], 1), {k: round(1e3 * v, 1) for k, v in d["kernel_ms"].items()})
        break
else:
    print(open(sys.argv[1]).read()[-1500:])
PY
done
