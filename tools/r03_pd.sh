#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python3 -m pytest tests -m gpu -x -q -k "left_looking or panel_orders or c5_nlml or g2_g8" 2>&1 | tail -2
python3 tools/c5_only.py chol_panel_diag=1 chol_panel_diag=0 chol_panel_diag=1,nlml_groups=1 chol_panel_diag=0,nlml_groups=1 2>&1 | tail -8
python3 tools/fuzz_nlml.py 40 5 | tail -1
