#!/bin/bash
# every launch of the last fit in start order with its queue: bash tools/fit_timeline.sh 4096 [key=value ...]
N=${1:-4096}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/fit_timeline; mkdir -p gpurun_out/fit_timeline
timeout -k 5 900 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fit_timeline/t -- python3 tools/time_fit.py "$@" $N > gpurun_out/fit_timeline/log.txt 2>&1
python3 - <<'PY' > gpurun_out/fit_timeline/timeline.txt
import csv, glob
f = glob.glob('gpurun_out/fit_timeline/t/*/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
last = max(i for i, r in enumerate(rows) if 'cov_matrix' in r['Kernel_Name'])
seq = rows[last:]
t0 = int(seq[0]['Start_Timestamp'])
print("span %.1f us, %d launches" % ((max(int(r['End_Timestamp']) for r in seq) - t0) / 1e3, len(seq)))
for r in seq:
    print("q%-3s %-34s start %8.1f  end %8.1f  dur %7.1f  grid %sx%s" % (r.get('Queue_Id', '?'), r['Kernel_Name'].split('(')[0][:34],
          (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3,
          r.get('Grid_Size_X', '?'), r.get('Grid_Size_Y', '?')))
PY
head -3 gpurun_out/fit_timeline/timeline.txt; tail -3 gpurun_out/fit_timeline/log.txt
