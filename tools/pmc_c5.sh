#!/bin/bash
# PMC passes over the C5 grid: instruction mix and MFMA busy of its kernels (kernels run one at a time under the counters)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_c5
rm -rf $OUT; mkdir -p $OUT
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_INSTS_MFMA" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $pass | cut -d" " -f1)
  timeout -k 5 900 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pmc_$tag -- python3 tools/c5_only.py > $OUT/pmc_$tag.log 2>&1
done
python3 - <<PY > $OUT/summary.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "update" in k or "rows" in k or "panel_" in k or "cov_grid" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    tot = {n: sum(v) for n, v in c.items()}
    print(k, "launches", len(c.get("SQ_INSTS_MFMA", [])))
    for n in sorted(tot): print("   %-28s %.4g" % (n, tot[n]))
    if tot.get("SQ_INSTS_MFMA") and "SQ_INSTS_VALU" in tot: print("   non-MFMA VALU per MFMA %.3f" % ((tot["SQ_INSTS_VALU"] - tot["SQ_INSTS_MFMA"]) / tot["SQ_INSTS_MFMA"]))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in tot: print("   MFMA busy %.1f %%" % (100 * tot["SQ_VALU_MFMA_BUSY_CYCLES"] / (tot["GRBM_GUI_ACTIVE"] / 8 * 1024)))
    if "FETCH_SIZE" in tot: print("   HBM-side GB (x2 fetch corr.): read %.2f write %.2f" % (2 * tot["FETCH_SIZE"] / 1e6 * 1.024, tot["WRITE_SIZE"] / 1e6 * 1.024))
PY
cat $OUT/summary.txt
