#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06j; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q --maxfail=8 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest.log
export ISCA_BENCH_NO_EXTRA=1
for rep in 1 2 3; do
  for V in exp:- flag:- exp:ISCA_NO_DEFERRED_FINISH=1; do
    L=${V%%:*}; E=${V#*:}
    ( export ISCA_DYN_LIB=$GRAFT_REPO_ROOT/isca_amd/lib/libisca_dyn_$L.so; [ "$E" != "-" ] && export $E
      timeout 300 python bench.py --workload T85L40 --steps 400 --warmup 40 --cpu-steps 0 > $OUT/b_${L}_${E//[^A-Za-z0-9_=]/_}_$rep.log 2>&1 )
    python - $OUT/b_${L}_${E//[^A-Za-z0-9_=]/_}_$rep.log "$V" <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    if ln.startswith("{"):
        d = json.loads(ln); print(sys.argv[2], "ms/step", round(d["ms_per_step"], 4), {k: round(1e3 * v, 1) for k, v in d["kernel_ms"].items()}); break
else:
    print(sys.argv[2], open(sys.argv[1]).read()[-600:])
PY
  done
done
