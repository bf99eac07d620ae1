#!/bin/bash
# kernel trace of maximizeEI (GPU box): the dependent chain of one batch -- kernel durations and the gaps between them
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/direct_trace; mkdir -p gpurun_out/direct_trace
timeout -k 5 900 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/direct_trace/t -- python3 tools/time_direct.py > gpurun_out/direct_trace/log.txt 2>&1
python3 - <<'PY' > gpurun_out/direct_trace/summary.txt
import csv, glob, collections
import numpy as np
f = glob.glob('gpurun_out/direct_trace/t/*/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'].split('(')[0].replace('void ', '')[:28] for r in rows]
# batches: kstar_small -> wk_small -> small_finish
i = 0; per = collections.defaultdict(list)
while i + 2 < len(rows):
    if names[i].startswith('kstar_small') and names[i + 1].startswith('wk_small') and names[i + 2].startswith('small_finish'):
        a, b, c = rows[i:i + 3]
        npad = int(b['Grid_Size_Y'])       # row blocks of W
        st = lambda r: int(r['Start_Timestamp']); en = lambda r: int(r['End_Timestamp'])
        nxt = st(rows[i + 3]) if i + 3 < len(rows) and names[i + 3].startswith('kstar_small') else None
        per[npad].append(((en(a) - st(a)) / 1e3, (st(b) - en(a)) / 1e3, (en(b) - st(b)) / 1e3, (st(c) - en(b)) / 1e3, (en(c) - st(c)) / 1e3,
                          (nxt - en(c)) / 1e3 if nxt else np.nan))
        i += 3
    else:
        i += 1
for npad, v in sorted(per.items()):
    v = np.array(v)
    print("W row-blocks %4d: %5d batches  kstar %.1f  gap %.1f  wk %.1f  gap %.1f  finish %.1f  -> next batch's first kernel %.1f us (medians)" %
          ((npad, len(v)) + tuple(np.nanmedian(v, axis=0))))
PY
cat gpurun_out/direct_trace/summary.txt; cat gpurun_out/direct_trace/log.txt
