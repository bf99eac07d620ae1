#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_left; mkdir -p $O; rm -rf $O/trace
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/c5_only.py nlml_groups=1 > $O/trace.log 2>&1
cp $O/trace/*/*_kernel_stats.csv $O/c5_left_kernel_stats.csv
python3 - <<'PY'
import csv
rows=list(csv.reader(open('gpurun_out/r03_left/c5_left_kernel_stats.csv')))
tot=sum(float(r[2]) for r in rows[1:])
for r in rows[1:9]: print("%-40s n=%5s total %8.2f ms avg %8.1f us %5.1f%%"%(r[0][:40], r[1], float(r[2])/1e6/4, float(r[3])/1e3, 100*float(r[2])/tot))
print(tot/4e6)
PY
