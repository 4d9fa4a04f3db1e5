cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for V in 0 141 142 181 182 121 111; do
  ISCA_LEG_INV=$V timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "golden_kernels or golden_run or T85L40 or T170L60 or restart" > gpurun_out/pt_$V.log 2>&1
  echo "inv variant $V pytest rc=$? $(tail -1 gpurun_out/pt_$V.log)"
done
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_full.log 2>&1; echo "full rc=$? $(tail -1 gpurun_out/pytest_full.log)"
export ISCA_TRACER_SERIAL=1
bash tools/leg_sweep.sh sweep4 T170L60 43:0 43:141 43:181 43:182 43:22
bash tools/leg_sweep.sh sweep4 T85L40 43:0 43:142 43:181 43:22
unset ISCA_TRACER_SERIAL
bash tools/leg_sweep.sh sweep4 "T170L60 T85L40" 43:0
