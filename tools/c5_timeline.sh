#!/bin/bash
# kernel timeline of ONE C5 grid (N = 4096, D = 16, 64 theta; GPU box): every launch of the last grid with its queue, start and duration,
# the union of busy time and per-kernel totals -- where the two sub-batches' chains overlap and where nothing runs
#   bash tools/c5_timeline.sh [out-dir]
O=${1:-gpurun_out/c5_timeline}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf $O; mkdir -p $O
timeout -k 5 900 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 tools/c5_only.py nlml_batch=0 > $O/log.txt 2>&1
python3 - $O <<'PY' > $O/timeline.txt
import csv, glob, sys, collections
O = sys.argv[1]
f = glob.glob(O + '/t/*/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last grid: from the last cov_grid_kernel pair on
cg = [i for i, r in enumerate(rows) if 'cov_grid' in r['Kernel_Name']]
if not cg: sys.exit('no cov_grid launch in the trace (did the run fail? see log.txt)')
first = cg[-2] if len(cg) >= 2 else cg[-1]
seq = rows[first:]
t0 = int(seq[0]['Start_Timestamp']); t1 = max(int(r['End_Timestamp']) for r in seq)
qk = 'Queue_Id' if 'Queue_Id' in seq[0] else ('Stream_Id' if 'Stream_Id' in seq[0] else None)
print("columns:", list(seq[0].keys()))
print("span %.1f us, %d launches" % ((t1 - t0) / 1e3, len(seq)))
ev = sorted([(int(r['Start_Timestamp']), 1) for r in seq] + [(int(r['End_Timestamp']), -1) for r in seq])
busy = 0; depth = 0; last = None; two = 0
for t, d in ev:
    if depth > 0: busy += t - last
    if depth > 1: two += t - last
    depth += d; last = t
print("some kernel running %.1f us (%.1f %%), two or more %.1f us" % (busy / 1e3, 100.0 * busy / (t1 - t0), two / 1e3))
tot = collections.OrderedDict()
for r in seq:
    n = r['Kernel_Name'].split('(')[0][:40]
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    a = tot.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += d
for n, (c, d) in tot.items(): print("%-42s x%4d  %9.1f us  (%.1f each)" % (n, c, d, d / c))
print("-- launches: queue, kernel, start us, duration us, grid")
for r in seq:
    print("%4s %-28s %9.1f %8.1f  %sx%sx%s" % (r.get(qk, '?') if qk else '?', r['Kernel_Name'].split('(')[0][:28], (int(r['Start_Timestamp']) - t0) / 1e3,
          (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r.get('Grid_Size_X', '?'), r.get('Grid_Size_Y', '?'), r.get('Grid_Size_Z', '?')))
PY
head -30 $O/timeline.txt
