#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_direct; mkdir -p $O; rm -rf $O/trace
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 tools/time_direct.py > $O/trace.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r03_direct/trace/*/*_kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# find the second maximizeEI at N=1024: sequences of kstar_small kernels; take batches 60..110
ks=[i for i,r in enumerate(rows) if 'kstar_small' in r['Kernel_Name'] or 'wkl_small' in r['Kernel_Name'] or 'wkf_small' in r['Kernel_Name']]
print(len(ks),"kstar launches")
sel=ks[60:66]+ks[170:176]+ks[270:276]
for i in sel:
    seq=rows[i:i+2]
    t0=int(seq[0]['Start_Timestamp'])
    prev_end=int(rows[i-1]['End_Timestamp'])
    s=" ".join("%s %5.1f(+%4.1f)"%(r['Kernel_Name'][:6],(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,(int(r['Start_Timestamp'])-t0)/1e3) for r in seq)
    print("gap since prev batch end %6.1f us | %s | grid %s"%((t0-prev_end)/1e3, s, seq[1]['Grid_Size_X']+"x"+seq[1].get('Grid_Size_Y','')))
PY
