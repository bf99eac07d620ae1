cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c5
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c5/trace -- python3 tools/run_configs.py c5 > gpurun_out/c5/trace.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/c5/trace/*/*_kernel_stats.csv')[0]
for r in csv.reader(open(f)): print(r[0][:40], r[1], r[2], r[3], r[5], r[6] if len(r)>6 else '')
PY
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c5/trace2 -- python3 tools/time_fit.py 4096 > gpurun_out/c5/trace2.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/c5/trace2/*/*_kernel_stats.csv')[0]
for r in csv.reader(open(f)): print(r[0][:40], r[1], r[2], r[3], r[5], r[6] if len(r)>6 else '')
PY
