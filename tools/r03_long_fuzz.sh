#!/bin/bash
# long randomised parity runs (GPU box): bash tools/r03_long_fuzz.sh
mkdir -p gpurun_out/long
FUZZ_DIRECT_CASES=40 timeout 2400 python3 tools/fuzz_gpu.py 1500 99 > gpurun_out/long/fuzz_gpu.txt 2>&1
{ grep -c "rel err" gpurun_out/long/fuzz_gpu.txt; grep "FAIL" gpurun_out/long/fuzz_gpu.txt | head; grep "worst" gpurun_out/long/fuzz_gpu.txt; tail -4 gpurun_out/long/fuzz_gpu.txt; } > gpurun_out/long/r03_fuzz_long_summary.txt
timeout 1200 python3 tools/fuzz_gallery.py 3000 21 > gpurun_out/long/fuzz_gallery.txt 2>&1
{ grep -c " ok:" gpurun_out/long/fuzz_gallery.txt; grep "FAIL" -A3 gpurun_out/long/fuzz_gallery.txt | head -12; tail -1 gpurun_out/long/fuzz_gallery.txt; } > gpurun_out/long/r03_fuzz_gallery_long_summary.txt
timeout 1200 python3 tools/fuzz_nlml.py 400 5 > gpurun_out/long/fuzz_nlml.txt 2>&1
tail -3 gpurun_out/long/fuzz_nlml.txt > gpurun_out/long/r03_fuzz_nlml_long_summary.txt
cat gpurun_out/long/r03_fuzz_long_summary.txt gpurun_out/long/r03_fuzz_gallery_long_summary.txt gpurun_out/long/r03_fuzz_nlml_long_summary.txt
