#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_direct; rm -rf $O; mkdir -p $O
timeout 900 python3 -m pytest tests -m gpu -x -q -k "direct or g1 or g3 or gallery or legacy or loop or smoke or small_batch or seventeen" > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt
for i in 1 2 3; do python3 tools/time_direct.py; done > $O/time_direct.txt 2>&1
IBO_DEBUG=1 python3 tools/time_direct.py > $O/time_direct_debug.txt 2>&1
tail -4 $O/pytest.txt; cat $O/time_direct.txt; grep DIRECT $O/time_direct_debug.txt | tail -6
