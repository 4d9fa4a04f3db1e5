#!/bin/bash
# A/B of this tree against an earlier commit inside ONE GPU-box visit (boxes differ by +-2 %): alternating bench runs of both trees.
#   git worktree add -f build/old <commit> && rm -rf build/old/tests/golden build/old/profiles && (cd build/old && python -m isca_amd.build)
#   gpurun -- 'bash tools/ab_old_new.sh'          (build/ is git-ignored and travels with the snapshot; remove the worktree afterwards)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab1
export ISCA_BENCH_NO_EXTRA=1
for rep in 1 2 3; do
  for W in T85L40 T170L60; do
    for T in new old; do
      D=$GRAFT_REPO_ROOT; [ $T = old ] && D=$GRAFT_REPO_ROOT/build/old
      ( cd $D && timeout 300 python bench.py --workload $W --steps 400 --warmup 40 --cpu-steps 0 > $GRAFT_REPO_ROOT/gpurun_out/ab1/${T}_${W}_$rep.log 2>&1 )
      python - $GRAFT_REPO_ROOT/gpurun_out/ab1/${T}_${W}_$rep.log $T $W $rep <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    if ln.startswith("{"):
        d = json.loads(ln)
        print(sys.argv[2], sys.argv[3], sys.argv[4], "ms/step", round(d["ms_per_step"], 4), "steady", round(d.get("steady_ms_per_step") or 0, 4), {k: round(1e3 * v, 1) for k, v in d["kernel_ms"].items()})
        break
else:
    print(sys.argv[2:], open(sys.argv[1]).read()[-800:])
PY
    done
  done
done
