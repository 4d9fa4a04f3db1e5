cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for pm in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $pm --output-format csv -d gpurun_out/pmc2_$i -o p -- python bench.py --steps 30 --warmup 10 --cpu-steps 0 > gpurun_out/pmc2_$i.log 2>&1
done
python - <<PY
import csv, glob, collections
tot = collections.defaultdict(dict)
for d in sorted(glob.glob("gpurun_out/pmc2_*/")):
    f = glob.glob(d + "*counter_collection.csv")
    if not f: print(d, "no counter file", open(d.rstrip("/")+".log").read()[-500:]); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f[0])):
        acc[row["Kernel_Name"].split("(")[0][-28:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in acc.items():
        for c, v in cs.items(): tot[k][c] = sum(v)/len(v)
for k in tot:
    if "k_leg" in k or "k_fft" in k or "k_column" in k or "spec_update" in k or "synth" in k:
        print(k, {c: round(v) for c, v in tot[k].items()})
PY
