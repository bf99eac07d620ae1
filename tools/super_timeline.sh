#!/bin/bash
# kernel timeline of the last fit of tools/time_fit.py (from its cov_fit_kernel to its alpha_reduce_kernel), per queue:
#   bash tools/super_timeline.sh 4096 [key=value ...]
N=${1:-4096}; shift
O=gpurun_out/super_timeline
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf $O; mkdir -p $O
timeout -k 5 600 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 tools/time_fit.py "$@" $N > $O/log.txt 2>&1
python3 - $O <<'PY' > $O/timeline.txt
import csv, glob, sys, collections
O = sys.argv[1]
f = glob.glob(O + '/t/*/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
cf = [i for i, r in enumerate(rows) if 'cov_fit_kernel' in r['Kernel_Name']]
if not cf: sys.exit("no cov_fit_kernel in the trace; see log.txt")
first = cf[-1]
end = next((i for i in range(first, len(rows)) if 'alpha_reduce' in rows[i]['Kernel_Name']), len(rows) - 1)
seq = rows[first:end + 1]
t0 = int(seq[0]['Start_Timestamp'])
print("span %.1f us, %d launches" % ((max(int(r['End_Timestamp']) for r in seq) - t0) / 1e3, len(seq)))
tot = collections.OrderedDict()
for r in seq:
    n = r['Kernel_Name'].split('(')[0][:36]
    a = tot.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for n, (c, d) in tot.items(): print("%-38s x%3d %9.1f us" % (n, c, d))
for r in seq:
    print("q%-3s %-30s start %8.1f  dur %7.1f  grid %s" % (r.get('Queue_Id', '?'), r['Kernel_Name'].split('(')[0][:30],
          (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r.get('Grid_Size_X', '?')))
PY
tail -2 $O/log.txt
