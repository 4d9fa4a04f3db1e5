cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_full.log 2>&1; echo "full rc=$? $(tail -1 gpurun_out/pytest_full.log)"
ISCA_TRACER_SERIAL=1 bash tools/leg_sweep.sh sweep5 "T170L60 T85L40" 0:0
bash tools/leg_sweep.sh sweep5 "T170L60 T85L40" 0:0
ISCA_TRACER_SERIAL=1 bash tools/leg_trace.sh "T85L40 T170L60" 2>&1 | grep -v "^  *[0-9.]* : " | awk '/what=/ {n++; if (n%6) next} {print}'
