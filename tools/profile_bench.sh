#!/bin/bash
# rocprofv3 passes over `python3 bench.py` (run on the GPU box from the repo root):
#   1. --kernel-trace --stats         -> per-kernel durations
#   2..4 --pmc (own runs, no other trace domains) -> MFMA utilisation, HBM bytes
# Raw output goes to gpurun_out/prof_$TAG/, summaries to profiles/${TAG}_*.
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT profiles
timeout -k 5 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $OUT/bench_trace.log 2>&1
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM" "SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $pass | cut -d" " -f1)
  timeout -k 5 900 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pmc_$tag -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_$tag.log 2>&1
done
python3 tools/summarize_profile.py $OUT $TAG
