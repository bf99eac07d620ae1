#!/bin/bash
# two quick rocprofv3 --pmc passes over the C2 bench (sweep kernel only): instruction mix and MFMA busy
TAG=${1:-q}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmcq_$TAG
mkdir -p $OUT
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_INSTS_MFMA" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM" "SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_WAVES"; do
  tag=$(echo $pass | cut -d" " -f1)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pmc_$tag -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_$tag.log 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "sweep2_kernel" in r["Kernel_Name"] or "sweep_mfma" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
c = {k: sum(v) / len(v) for k, v in acc.items()}
for k in sorted(c): print("%-28s %.4g" % (k, c[k]))
if "SQ_INSTS_MFMA" in c and "SQ_INSTS_VALU" in c:
    print("non-MFMA VALU per MFMA: %.3f" % ((c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / c["SQ_INSTS_MFMA"]))
if "SQ_VALU_MFMA_BUSY_CYCLES" in c: print("MFMA busy %% : %.2f" % (100 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024)))
PY
