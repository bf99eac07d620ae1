cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/direct
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/direct/trace -- python3 tools/time_direct.py > gpurun_out/direct/trace.log 2>&1
python3 - <<'PY'
import csv,glob,os
f=max(glob.glob('gpurun_out/direct/trace/*/*_kernel_stats.csv'), key=os.path.getmtime)
for r in list(csv.reader(open(f)))[:12]: print(r[0][:60], r[1], r[3], r[5], r[6])
PY
tail -3 gpurun_out/direct/trace.log
