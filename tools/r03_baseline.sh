#!/bin/bash
# round-3 baseline: C5 per-kernel stats, fit traces, short bench (GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_base; rm -rf $O; mkdir -p $O
bash tools/run_c5_profile.sh > $O/c5_profile.txt 2>&1
cp gpurun_out/c5/trace/*/*_kernel_stats.csv $O/c5_kernel_stats.csv
for n in 2048 4096; do bash tools/fit_trace.sh $n > $O/fit_trace_$n.txt 2>&1; done
python3 tools/c5_only.py > $O/c5_only.txt 2>&1
python3 bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err
tail -3 $O/c5_only.txt; head -40 $O/c5_profile.txt
