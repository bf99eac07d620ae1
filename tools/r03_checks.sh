#!/bin/bash
# round-3 correctness pass on the GPU box: full -m gpu suite, tolerance probe, launcher path
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_checks; rm -rf $O; mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
timeout 900 python3 tools/tolerance_probe.py $O/tolerance_probe.txt > /dev/null 2> $O/tolerance_probe.err
timeout 1200 python3 tools/launcher_on_one_gpu.py $O/launcher_on_one_gpu.json > $O/launcher.log 2>&1; echo "launcher rc=$?" >> $O/launcher.log
tail -5 $O/pytest_gpu.txt; tail -8 $O/tolerance_probe.txt; tail -5 $O/launcher.log
