#!/bin/bash
# per-launch durations of the pipelined block columns of one N-point fit (GPU box):  bash tools/pipe_columns.sh 4096 [option=value ...]
N=${1:-4096}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pipe_cols; mkdir -p gpurun_out/pipe_cols
timeout -k 5 900 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pipe_cols/t -- python3 tools/time_fit.py $N "$@" > gpurun_out/pipe_cols/log.txt 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/pipe_cols/t/*/*_kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
lastp = max(i for i, r in enumerate(rows) if 'chol_pipe8' in r['Kernel_Name'])
last = max(i for i, r in enumerate(rows[:lastp]) if 'cov_fit' in r['Kernel_Name'] or 'cov_matrix' in r['Kernel_Name'])      # the last fit's covariance pass (a later one is GP.R on request)
seq = [r for r in rows[last:] if 'chol_pipe8' in r['Kernel_Name']]
t0 = int(rows[last]['Start_Timestamp'])
print("column: grid (workgroups), duration us, gap to the previous launch us")
prev = None
for j, r in enumerate(seq):
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print("%3d  %6s  %7.1f  %6.1f" % (j, int(r['Grid_Size_X']) // 512, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0))
    prev = e
print("sum of durations %.1f us, span %.1f us" % (sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seq) / 1e3,
      (int(seq[-1]['End_Timestamp']) - int(seq[0]['Start_Timestamp'])) / 1e3))
PY
