#!/bin/bash
# long randomised parity runs for the end of a round (GPU box, from the repo root):  bash tools/long_fuzz.sh rNN
#   fit + posterior + EI sweeps and DIRECT runs against the oracle; kept-state (gallery) sweeps pruned / lazy against everything-completed (bit for bit)
#   and the one-kernel state; NLML grids left- against right-looking (bit for bit) and against the oracle
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/longfuzz_$TAG; mkdir -p $O profiles
FUZZ_DIRECT_CASES=40 timeout 2400 python3 tools/fuzz_gpu.py 1500 41 > $O/fuzz.txt 2>&1
{ grep -c "rel err" $O/fuzz.txt; grep "FAIL" $O/fuzz.txt | head; grep "worst" $O/fuzz.txt; tail -4 $O/fuzz.txt; } > profiles/${TAG}_fuzz_long_summary.txt
timeout 1800 python3 tools/fuzz_gallery.py 3000 43 > $O/fuzz_gallery.txt 2>&1
{ grep -c " ok:" $O/fuzz_gallery.txt; grep "FAIL" -A3 $O/fuzz_gallery.txt | head -20; tail -1 $O/fuzz_gallery.txt; } > profiles/${TAG}_fuzz_gallery_long_summary.txt
timeout 1800 python3 tools/fuzz_nlml.py 400 47 > $O/fuzz_nlml.txt 2>&1
tail -4 $O/fuzz_nlml.txt > profiles/${TAG}_fuzz_nlml_long_summary.txt
# the round-6 routes under the same oracle: fits from 2048 rows in super-panels (sizes up to 2600), DIRECT's batches on the resident evaluation server
FUZZ_OPTS=super_min_nb=32,direct_resident=1 FUZZ_NMAX=2600 FUZZ_DMAX=16 FUZZ_DIRECT_CASES=60 timeout 2400 python3 tools/fuzz_gpu.py 260 53 > $O/fuzz_routes.txt 2>&1
{ grep -c "rel err" $O/fuzz_routes.txt; grep -c "N= 2[0-9][0-9][0-9]" $O/fuzz_routes.txt; grep "FAIL" $O/fuzz_routes.txt | head; grep "worst" $O/fuzz_routes.txt; tail -1 $O/fuzz_routes.txt; } > profiles/${TAG}_fuzz_routes_long_summary.txt
mkdir -p $O/profiles; cp profiles/${TAG}_fuzz*long* $O/profiles/
cat profiles/${TAG}_fuzz_long_summary.txt profiles/${TAG}_fuzz_gallery_long_summary.txt profiles/${TAG}_fuzz_nlml_long_summary.txt profiles/${TAG}_fuzz_routes_long_summary.txt
