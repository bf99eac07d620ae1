#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_groups; rm -rf $O; mkdir -p $O
timeout 300 python3 tools/c5_only.py chol_left=1,nlml_groups=1 chol_left=1,nlml_groups=2 chol_left=1,nlml_groups=3 chol_left=1,nlml_groups=4 chol_left=0,nlml_groups=1 > $O/c5_ab.txt 2>&1
cat $O/c5_ab.txt
