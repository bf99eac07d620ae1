cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/fit
python3 tools/time_fit.py > gpurun_out/fit/time.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fit/trace -- python3 tools/time_fit.py 1024 > gpurun_out/fit/trace.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/fit/trace/*/*_kernel_stats.csv')[0]
for r in csv.reader(open(f)): print(r[0][:40], r[1], r[3], r[5])
PY
cat gpurun_out/fit/time.log
python3 -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
