#!/bin/bash
# kernel trace of one N-point fit (GPU box): per-kernel totals and the launch sequence of the last fit
# bash tools/fit_trace.sh 4096
N=${1:-4096}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/fit_trace; mkdir -p gpurun_out/fit_trace
timeout -k 5 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fit_trace/t -- python3 tools/time_fit.py $N > gpurun_out/fit_trace/log.txt 2>&1
python3 - <<'PY' > gpurun_out/fit_trace/summary.txt
import csv, glob, collections
f = glob.glob('gpurun_out/fit_trace/t/*/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last fit: from the last cov_matrix_kernel on
lastc = max(i for i, r in enumerate(rows) if 'chol_' in r['Kernel_Name'])
last = max(i for i, r in enumerate(rows[:lastc]) if 'cov_fit' in r['Kernel_Name'] or 'cov_matrix' in r['Kernel_Name'])      # the last fit's covariance pass (a later one is GP.R, formed on request)
seq = [r for i, r in enumerate(rows[last:]) if i == 0 or 'cov_matrix' not in r['Kernel_Name']]
t0 = int(seq[0]['Start_Timestamp'])
tot = collections.OrderedDict()
for r in seq:
    n = r['Kernel_Name'].split('(')[0][:48]
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    a = tot.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += d
print("span %.1f us, %d launches" % ((int(seq[-1]['End_Timestamp']) - t0) / 1e3, len(seq)))
for n, (c, d) in tot.items(): print("%-50s x%4d  %9.1f us  (%.1f each)" % (n, c, d, d / c))
print("-- trinv launches in order")
for r in seq:
    if 'trinv' in r['Kernel_Name']:
        print("%-40s start %8.1f us  dur %7.1f us  grid %s" % (r['Kernel_Name'].split('(')[0][:40], (int(r['Start_Timestamp']) - t0) / 1e3,
              (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r.get('Grid_Size_X', '?') + "x" + r.get('Grid_Size_Y', '?')))
PY
cat gpurun_out/fit_trace/summary.txt
