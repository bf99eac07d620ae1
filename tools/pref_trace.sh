#!/bin/bash
# kernel trace of one PrefGaussianProcess construction (addPreferences: Newton MAP + the final factorisation) at P pairs, D = 6 (GPU box):
#   bash tools/pref_trace.sh 512
P=${1:-512}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pref_trace; mkdir -p gpurun_out/pref_trace
cat > gpurun_out/pref_trace/run.py <<PY
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from ibo_amd.gaussianprocess import PrefGaussianProcess
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
def hartman6(x):
    A = np.array([[10, 3, 17, 3.5, 1.7, 8], [0.05, 10, 17, 0.1, 8, 14], [3, 3.5, 1.7, 10, 17, 8], [17, 8, 0.05, 10, 0.1, 14]])
    Pm = np.array([[0.1312, 0.1696, 0.5569, 0.0124, 0.8283, 0.5886], [0.2329, 0.4135, 0.8307, 0.3736, 0.1004, 0.9991],
                   [0.2348, 0.1451, 0.3522, 0.2883, 0.3047, 0.6650], [0.4047, 0.8828, 0.8732, 0.5743, 0.1091, 0.0381]])
    C = np.array([1, 1.2, 3, 3.2])
    return float(np.sum(C * np.exp(-np.sum(A * (x - Pm) ** 2, axis=1))))
P = $P
rs = np.random.RandomState(4)
pts = rs.rand(2 * P, 6)
prefs = []
for i in range(P):
    a, b = pts[2 * i], pts[2 * i + 1]
    prefs.append((a, b, 0) if hartman6(a) > hartman6(b) else (b, a, 0))
for _ in range(3):
    t0 = time.perf_counter()
    GP = PrefGaussianProcess(GaussianKernel_ard([0.53, 0.57, 2.5, 0.34, 0.27, 0.35]), prefs)
    print("addPreferences %.2f ms" % ((time.perf_counter() - t0) * 1e3))
PY
timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pref_trace/t -- python3 gpurun_out/pref_trace/run.py > gpurun_out/pref_trace/log.txt 2>&1
grep addPref gpurun_out/pref_trace/log.txt
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/pref_trace/t/*/*_kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
n = len(rows) // 3
seq = rows[2 * n:]                                   # the last of three identical constructions
t0 = int(seq[0]['Start_Timestamp'])
print("last construction: span %.1f us, %d launches, kernel time %.1f us" % ((int(seq[-1]['End_Timestamp']) - t0) / 1e3, len(seq),
      sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seq) / 1e3))
tot = collections.OrderedDict()
for r in seq:
    k = r['Kernel_Name'].split('(')[0][:56]
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    a = tot.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += d
for k, (c, d) in sorted(tot.items(), key=lambda kv: -kv[1][1]): print("%-58s x%4d  %9.1f us  (%.1f each)" % (k, c, d, d / c))
PY
