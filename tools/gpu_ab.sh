#!/bin/bash
# One GPU-box visit: the -m gpu tests, then short bench runs (kernel timings included) of the headline workload under a list of
# environment variants ("NAME=VALUE,NAME=VALUE" per variant; "-" = defaults), each run twice, and of T170L60 once.
#   usage (through gpurun): bash tools/gpu_ab.sh <tag> "<pytest args>" variant [variant...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-ab}; PYARGS=${2:-}; shift; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export ISCA_BENCH_NO_EXTRA=1
if [ "$PYARGS" != "skip" ]; then
  timeout 1200 python -m pytest tests -m gpu -x -q $PYARGS > $OUT/pytest.log 2>&1
  echo "pytest rc=$?"; tail -6 $OUT/pytest.log
fi
show() {
  python - "$1" "$2" <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    if ln.startswith("{"):
        d = json.loads(ln)
        print(sys.argv[2], d["config"]["workload"][:8], "ms/step", round(d["ms_per_step"], 4), "SYPD", round(d["value"], 1), {k: round(1e3 * v, 1) for k, v in d["kernel_ms"].items()})
        break
else:
    print(sys.argv[2], open(sys.argv[1]).read()[-1500:])
PY
}
for V in "$@"; do
  for rep in 1 2; do
    ( [ "$V" != "-" ] && export ${V//,/ }; W=${ISCA_AB_WORKLOAD:-T85L40}
      timeout 300 python bench.py --workload $W --steps 300 --warmup 40 --cpu-steps 0 > $OUT/bench_${V//[^A-Za-z0-9_=]/_}_$rep.log 2>&1 )
    show $OUT/bench_${V//[^A-Za-z0-9_=]/_}_$rep.log "$V#$rep"
  done
done
