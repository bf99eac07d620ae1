#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python3 -m pytest tests -m gpu -x -q -k "left_looking or panel_orders or c5_nlml or g2_g8" 2>&1 | tail -2
python3 tools/c5_only.py chol_tail=16 chol_tail=0 chol_tail=8 chol_tail=12 chol_tail=20 chol_tail=24 chol_tail=32 2>&1 | tail -14
python3 tools/fuzz_nlml.py 60 7 | tail -1
