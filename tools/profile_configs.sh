#!/bin/bash
# rocprofv3 evidence for every roofline block of the bench line that is not the headline (GPU box, from the repo root):
#   bash tools/profile_configs.sh r03
# For each workload: one --kernel-trace --stats pass (per-kernel durations -> profiles/<tag>_<name>_kernel_stats.csv) and
# --pmc passes in their own runs (MFMA busy / instruction mix / FETCH_SIZE / WRITE_SIZE) condensed by
# tools/summarize_configs.py into profiles/<tag>_<name>_pmc.json, keyed by the hash of the kernel sources
# (bench.py only quotes a summary taken from the sources in the tree).
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/cfgprof_$TAG
rm -rf $OUT; mkdir -p $OUT profiles
P1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"
P2="FETCH_SIZE"
P3="WRITE_SIZE"
P4="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM"
run() {   # name, then the program and its arguments (the program itself after --: no shell in between)
  name=$1; shift
  timeout -k 5 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name/trace -- "$@" > $OUT/$name.trace.log 2>&1
  i=0
  for pass in "$P1" "$P2" "$P3" "$P4"; do
    i=$((i+1))
    timeout -k 5 900 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/$name/pmc$i -- "$@" > $OUT/$name.pmc$i.log 2>&1
  done
}
run c3 python3 bench.py --config c3 --steps 2 --warmup 1 --no-cpu-baseline
run c5 python3 tools/c5_only.py
run fit4096 python3 tools/time_fit.py 4096
run fit2048 python3 tools/time_fit.py 2048
run fit1024 python3 tools/time_fit.py 1024
run learn4096 python3 tools/learn_only.py 4096 16 8
run learn1024 python3 tools/learn_only.py 1024 16 8
# kernel stats only
timeout -k 5 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/gallery/trace -- python3 tools/run_configs.py c3 > $OUT/gallery.trace.log 2>&1
timeout -k 5 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4/trace -- python3 tools/run_configs.py c4 > $OUT/c4.trace.log 2>&1
python3 tools/summarize_configs.py $OUT $TAG
