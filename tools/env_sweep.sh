#!/bin/bash
# Kernel durations inside the step (rocprofv3 --kernel-trace --stats) for environment-switch variants of the library, on the GPU box.
# usage: bash tools/env_sweep.sh TAG "WORKLOADS" VARIANT...    VARIANT = "A=1,B=2" (comma-separated assignments) or "-" for none
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1; WL=$2; shift 2; mkdir -p $OUT
export ISCA_BENCH_NO_EXTRA=1
for W in $WL; do
  for V in "$@"; do
    D=$OUT/${W}_$(echo "$V" | tr -c 'A-Za-z0-9_\n' '_')
    ENVS=""; [ "$V" != "-" ] && ENVS=$(echo "$V" | tr ',' ' ')
    env $ENVS timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o r -- python bench.py --workload $W --steps 60 --warmup 10 --cpu-steps 0 > $D.log 2>&1
    python - "$W [$V]" $D $D.log <<'PY'
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
        for key in ("k_leg_fwd", "k_leg_inv", "k_fft_fwd", "k_fft_inv", "k_column", "k_spec_update", "k_tracer_horiz", "k_tracer_vert", "k_fixer_sums", "k_fixer_apply", "k_moist_physics"):
            if key in n and int(r["Calls"]) >= 50: out.append("%s %.1f" % (key[2:], float(r["AverageNs"]) / 1e3))
print(tag, "ms/step(under rocprof)", None if ms is None else round(ms, 4), "|", " ".join(sorted(out)))
PY
    rm -rf $D
  done
done
