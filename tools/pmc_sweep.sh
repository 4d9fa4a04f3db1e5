# usage: bash tools/pmc_sweep.sh [bench args]  -- per-kernel PMC counters of the bench (one counter group per run)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for pm in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $pm --output-format csv -d gpurun_out/pmc3_$i -o p -- python bench.py --steps 30 --warmup 10 --cpu-steps 0 "$@" > gpurun_out/pmc3_$i.log 2>&1
done
python - <<PY
import csv, glob, collections
tot = collections.defaultdict(dict)
for d in sorted(glob.glob("gpurun_out/pmc3_*/")):
    f = glob.glob(d + "*counter_collection.csv")
    if not f: print(d, "no counter file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f[0])):
        acc[row["Kernel_Name"].split("(")[0][-28:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in acc.items():
        for c, v in cs.items(): tot[k][c] = sum(v)/len(v)
for k in tot:
    if "isca" in k and tot[k].get("SQ_WAVES", 0) > 100:
        print(k, {c: round(v) for c, v in tot[k].items()})
PY
