#!/bin/bash
# Gaps between consecutive kernels of the step (end of one -> start of the next, per stream) from a rocprofv3 kernel trace of the bench:
#   gpurun -- 'bash tools/kernel_gaps.sh [workload]'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
W=${1:-T85L40}
OUT=gpurun_out/gaps; rm -rf $OUT; mkdir -p $OUT
export ISCA_BENCH_NO_EXTRA=1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o t -- python bench.py --workload $W --steps 200 --warmup 20 --cpu-steps 0 > $OUT/bench.log 2>&1
python - $OUT <<'PY'
import sys, glob, csv, collections
f = glob.glob(sys.argv[1] + "/tr/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].split("<")[0].replace("void isca::", "")
rows = rows[len(rows) // 2:]                      # the timed part
byq = collections.defaultdict(list)
for r in rows:
    byq[r["Queue_Id"]].append(r)
for q, rs in byq.items():
    gaps = collections.defaultdict(list); dur = collections.defaultdict(list)
    for a, b in zip(rs, rs[1:]):
        gaps[(name(a), name(b))].append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3)
    for r in rs:
        dur[name(r)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("queue", q, "kernels", len(rs))
    for k, v in sorted(gaps.items(), key=lambda x: -len(x[1]))[:14]:
        if len(v) > 20:
            v.sort(); print("  gap %-22s -> %-22s n=%4d median %6.2f us  mean %6.2f" % (k[0][:22], k[1][:22], len(v), v[len(v) // 2], sum(v) / len(v)))
    for k, v in sorted(dur.items(), key=lambda x: -sum(x[1]))[:12]:
        print("  dur %-24s n=%4d mean %7.2f us" % (k[:24], len(v), sum(v) / len(v)))
PY
rm -rf $OUT/tr
