#!/bin/bash
# rocprofv3 kernel stats of one fit size: bash tools/fit_trace.sh 4096
N=${1:-4096}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/fit_trace_$N
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fit_trace_$N -- python3 tools/time_fit.py $N > gpurun_out/fit_trace_$N.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('gpurun_out/fit_trace_$N/*/*_kernel_stats.csv')[0]
rows=list(csv.reader(open(f)))
for r in rows[:14]: print("%-46s calls %6s  avg %10s ns  total %12s  %s%%" % (r[0][:46], r[1], r[3][:10], r[2], r[4]))
PY
tail -2 gpurun_out/fit_trace_$N.log
