cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export ISCA_BENCH_NO_EXTRA=1
OUT=gpurun_out/pmc_dx; rm -rf $OUT; mkdir -p $OUT
for W in T170L60 T85L40; do
for pm in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $pm --output-format csv -d $OUT/${W}_$pm -o p -- python bench.py --workload $W --steps 30 --warmup 10 --cpu-steps 0 > $OUT/${W}_$pm.log 2>&1
done
python - $OUT $W <<'PY'
import sys, glob, csv, collections
out, W = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(lambda: [0.0, 0.0, 0, 0])
for pm in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{out}/{W}_{pm}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void isca::", "")
            v = float(r["Counter_Value"])
            t = tot[k]
            if pm == "FETCH_SIZE": t[0] += v; t[2] += 1
            else: t[1] += v; t[3] += 1
for k, (fe, wr, n, nw) in sorted(tot.items(), key=lambda x: -x[1][0]):       # (each pass averaged over its own dispatches: the bench's spin-up makes their numbers differ)
    if n and nw: print(W, k, "MB per launch (2*FETCH+WRITE KiB):", round((2 * fe / n + wr / nw) * 1024 / 1e6, 1), "fetch", round(2 * fe * 1024 / n / 1e6, 1), "write", round(wr * 1024 / nw / 1e6, 1))
PY
done
rm -rf $OUT/*_FETCH_SIZE $OUT/*_WRITE_SIZE
