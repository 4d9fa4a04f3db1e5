cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_full.log 2>&1; echo "full rc=$? $(grep -E 'passed|failed' gpurun_out/pytest_full.log | tail -1)"
