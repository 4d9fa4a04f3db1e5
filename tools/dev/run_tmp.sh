cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for V in "64 1" "32 2" "16 4" "32 4" "64 2" "16 2"; do
  set -- $V
  echo "== NC=$1 W=$2"
  ISCA_MOIST_NC=$1 ISCA_MOIST_W=$2 timeout 120 python tools/dev/moist_bench.py 2>&1 | grep -E "ms/step|moist_physics|moist_pressures" | tail -3
done
ISCA_MOIST_NC=32 ISCA_MOIST_W=2 timeout 300 python -m pytest tests/test_gpu_moist.py -x -q 2>&1 | tail -2
echo "== T170L60 moist"
for V in "64 1" "32 2" "16 4"; do set -- $V; ISCA_MOIST_NC=$1 ISCA_MOIST_W=$2 timeout 200 python tools/dev/moist_bench.py T170 60 150 2>&1 | grep -E "ms/step|moist_physics" | tail -2; done
