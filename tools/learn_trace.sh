#!/bin/bash
# kernel trace of one NLML + gradient evaluation (ibo_nlml_grad) at N points, D dimensions (GPU box):  bash tools/learn_trace.sh 1024 4
N=${1:-1024}; D=${2:-4}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/learn_trace; mkdir -p gpurun_out/learn_trace
cat > gpurun_out/learn_trace/run.py <<PY
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
from ibo_amd.gaussianprocess.trainhyper import marginalLikelihood
N, D = $N, $D
rs = np.random.RandomState(3)
X = rs.rand(N, D); Y = np.sin(3 * X.sum(1)) + .05 * rs.randn(N)
k = GaussianKernel_ard(np.full(D, .5))
for _ in range(4): marginalLikelihood(k, X, Y, D, True)
PY
timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/learn_trace/t -- python3 gpurun_out/learn_trace/run.py > gpurun_out/learn_trace/log.txt 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/learn_trace/t/*/*_kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if 'cov_matrix' in r['Kernel_Name'] or 'cov_fit' in r['Kernel_Name']]
seq = rows[starts[-1]:]
t0 = int(seq[0]['Start_Timestamp'])
print("last evaluation: span %.1f us, %d launches" % ((int(seq[-1]['End_Timestamp']) - t0) / 1e3, len(seq)))
tot = collections.OrderedDict()
for r in seq:
    n = r['Kernel_Name'].split('(')[0][:56]
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    a = tot.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += d
for n, (c, d) in tot.items(): print("%-58s x%4d  %9.1f us  (%.1f each)" % (n, c, d, d / c))
PY
