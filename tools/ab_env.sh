#!/bin/bash
# A/B of environment variants inside one GPU-box visit: bash tools/ab_env.sh <tag> <workload list> <reps> variant [variant...]   ("-" = defaults; NAME=VALUE,NAME=VALUE)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; WL=$2; REPS=$3; shift; shift; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export ISCA_BENCH_NO_EXTRA=1
for rep in $(seq 1 $REPS); do
  for W in $WL; do
    for V in "$@"; do
      ( [ "$V" != "-" ] && export ${V//,/ }
        timeout 300 python bench.py --workload $W --steps 400 --warmup 40 --cpu-steps 0 > $OUT/${W}_${V//[^A-Za-z0-9_=]/_}_$rep.log 2>&1 )
      python - $OUT/${W}_${V//[^A-Za-z0-9_=]/_}_$rep.log "$V" $W $rep <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    if ln.startswith("{"):
        d = json.loads(ln)
        print(sys.argv[3], sys.argv[2], sys.argv[4], "ms/step", round(d["ms_per_step"], 4), "steady", round(d.get("steady_ms_per_step") or 0, 4), {k: round(1e3 * v, 1) for k, v in d["kernel_ms"].items()})
        break
else:
    print(sys.argv[2:], open(sys.argv[1]).read()[-800:])
PY
    done
  done
done
