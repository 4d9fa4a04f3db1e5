#!/bin/bash
# Legendre kernel variants (ISCA_LEG_FWD = FD*10+NTG, ISCA_LEG_INV = JTG*10+ID: measurement switches of legendre.hip) on the GPU box:
# rocprofv3 kernel durations inside the step.   usage: bash tools/leg_sweep.sh TAG "WORKLOADS" FWD:INV FWD:INV ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1; WL=$2; shift 2; mkdir -p $OUT
export ISCA_BENCH_NO_EXTRA=1
for W in $WL; do
  for V in "$@"; do
    F=${V%%:*}; I=${V##*:}
    D=$OUT/${W}_${F}_${I}
    ISCA_LEG_FWD=$F ISCA_LEG_INV=$I timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o r -- python bench.py --workload $W --steps 60 --warmup 10 --cpu-steps 0 > $D.log 2>&1
    python - "$W fwd=$F inv=$I ${ISCA_TRACER_SERIAL:+serial}" $D $D.log <<'PY'
import csv, glob, json, sys
tag, d, log = sys.argv[1:4]
ms = None
for ln in open(log):
    if ln.startswith("{"): ms = json.loads(ln)["ms_per_step"]
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
out = []
if f:
    for r in csv.DictReader(open(f[0])):
        n = r["Name"]
        for key in ("k_leg_fwd", "k_leg_inv", "k_fft_fwd", "k_fft_inv", "k_column", "k_spec_update", "k_tracer_horiz", "k_tracer_vert", "k_fixer_sums", "k_fixer_apply"):
            if key in n and int(r["Calls"]) >= 50: out.append("%s %.1f" % (key[2:], float(r["AverageNs"]) / 1e3))
print(tag, "ms/step(under rocprof)", None if ms is None else round(ms, 4), "|", " ".join(sorted(out)))
PY
    rm -rf $D
  done
done
