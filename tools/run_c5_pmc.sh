cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c5pmc
i=0
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d gpurun_out/c5pmc/p$i -- python3 tools/run_configs.py c5 > gpurun_out/c5pmc/p$i.log 2>&1
done
python3 - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/c5pmc/p*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:24]
        if 'chol_update' in k and int(r['Grid_Size'])>=1000*256:
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in acc.items():
    print(k)
    for c,v in sorted(d.items()): print("   %-32s mean %.4g  max %.4g  n=%d" % (c, sum(v)/len(v), max(v), len(v)))
PY
