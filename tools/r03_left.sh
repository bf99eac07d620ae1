#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_left; rm -rf $O; mkdir -p $O
timeout 900 python3 -m pytest tests -m gpu -x -q -k "left_looking or panel_orders or c5_nlml or g2_g8 or learn_hyper" > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt
timeout 300 python3 tools/c5_only.py chol_panel_rows=3 chol_panel_rows=2 chol_panel_rows=1 chol_left=0,cov_fast=0,nlml_groups=1,chol_panel_rows=1 > $O/c5_ab.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/c5_only.py chol_left=1,nlml_groups=1 > $O/trace.log 2>&1
cp $O/trace/*/*_kernel_stats.csv $O/c5_left_kernel_stats.csv
tail -4 $O/pytest.txt; cat $O/c5_ab.txt
python3 - <<'PY'
import csv,glob
rows=list(csv.reader(open('gpurun_out/r03_left/c5_left_kernel_stats.csv')))
tot=sum(float(r[2]) for r in rows[1:])
for r in rows[1:11]: print("%-40s n=%5s total %8.2f ms avg %8.1f us %5.1f%%"%(r[0][:40], r[1], float(r[2])/1e6/4, float(r[3])/1e3, 100*float(r[2])/tot))
print(tot/4e6)
f=glob.glob('gpurun_out/r03_left/trace/*/*_kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f)) if 'update3' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
tot=0;tid=0
for p,r in enumerate(rows[-15:],1):
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    K=256*p
    useful=64*((4096-K+1)*256 - 128*256)*K*2
    ideal=useful/78.6e12*1e6
    tot+=d; tid+=ideal
    print("p=%2d K=%4d %8.1f us  ideal %7.1f  eff %.2f"%(p,K,d,ideal,ideal/d))
print(tot,tid,tid/tot)
PY
