cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_full.log 2>&1; echo "full rc=$? $(grep -E 'passed|failed' gpurun_out/pytest_full.log | tail -1)"
ISCA_BENCH_NO_EXTRA=1 timeout 300 python bench.py --steps 300 --warmup 50 --cpu-steps 0 > gpurun_out/bench_T85.log 2>&1; tail -c 1500 gpurun_out/bench_T85.log
