#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python3 -m pytest tests -m gpu -x -q -k "left_looking or panel_orders or c5_nlml or g2_g8 or two_level or split_steps or trinv" 2>&1 | tail -3
python3 tools/c5_only.py chol_left=1 chol_left=0 2>&1 | tail -4
for mt in 1024 100 300; do python3 tools/time_fit.py update2_min_tiles=$mt 4096 3000 8192; done
timeout 600 python3 tools/fuzz_nlml.py 50 2 2>&1 | tail -1
