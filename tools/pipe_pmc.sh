#!/bin/bash
# memory-side counters (each pass under its own timeout: a counter set the hardware cannot collect makes rocprofv3 abort and then hang)
# memory-side counters of the pipelined block columns of an N-point fit (GPU box):  bash tools/pipe_pmc.sh 4096 [option=value ...]
N=${1:-4096}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pipe_pmc; rm -rf $OUT; mkdir -p $OUT
i=0
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/p$i -- python3 tools/time_fit.py $N "$@" > $OUT/p$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(float); n = collections.defaultdict(int); dur = 0.0; nd = 0
for f in sorted(glob.glob('gpurun_out/pipe_pmc/p*/*/*_counter_collection.csv')):
    seen = set()
    for r in csv.DictReader(open(f)):
        if 'chol_pipe8' in r['Kernel_Name']:
            acc[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
            if r['Counter_Name'] == 'GRBM_GUI_ACTIVE': dur += int(r['End_Timestamp']) - int(r['Start_Timestamp']); nd += 1
for k in sorted(acc): print("%-40s %16.0f over %d launches (%.1f per fit of 8)" % (k, acc[k], n[k], acc[k] / 8))
if nd: print("kernel time %.3f ms per fit" % (dur / 1e6 / 8))
PY
tail -3 $OUT/p*.log | grep -i "error\|invalid\|not" | head
