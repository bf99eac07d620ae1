#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_left; rm -rf $O; mkdir -p $O
timeout 900 python3 -m pytest tests -m gpu -x -q -k "left_looking or panel_orders or c5_nlml or g2_g8 or alias or tolerance" > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt
timeout 300 python3 tools/c5_only.py chol_left=1 chol_left=0 > $O/c5_ab.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/c5_only.py chol_left=1 > $O/trace.log 2>&1
cp $O/trace/*/*_kernel_stats.csv $O/c5_left_kernel_stats.csv
tail -15 $O/pytest.txt; cat $O/c5_ab.txt; head -12 $O/c5_left_kernel_stats.csv | cut -c1-150
bash tools/pmc_c5.sh > $O/pmc_c5.txt 2>&1; tail -40 $O/pmc_c5.txt
