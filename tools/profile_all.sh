#!/bin/bash
# every rNN artefact under profiles/ from one GPU-box session (run from the repo root):  bash tools/profile_all.sh rNN
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/final_$TAG profiles
O=gpurun_out/final_$TAG
bash tools/profile_bench.sh $TAG > $O/profile_bench.log 2>&1
bash tools/profile_configs.sh $TAG > $O/profile_configs.log 2>&1
python3 bench.py --steps 20 --warmup 3 > profiles/${TAG}_bench_n1.json 2> $O/bench.err
python3 bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline > profiles/${TAG}_bench_c3_n1.json 2> $O/bench_c3.err
python3 bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline > profiles/${TAG}_bench_c5_n1.json 2> $O/bench_c5.err
python3 tools/run_configs.py c2 c3 c4 c5 > profiles/${TAG}_configs_c2_c5.jsonl 2> $O/configs.err
{ python3 tools/time_fit.py; python3 tools/time_direct.py; python3 tools/time_single_calls.py; python3 tools/time_learn.py; } > profiles/${TAG}_latencies.txt 2> $O/lat.err
python3 tools/time_incremental.py > profiles/${TAG}_time_incremental.txt 2> $O/inc.err
{ for c in c2 c3 c4; do for l in 4 3 2; do python3 tools/argmax_only.py $c $l; done; done; } > profiles/${TAG}_argmax_only.txt 2> $O/argmax.err
{ python3 tools/legacy_probe.py 1e-4 1 2>&1 | grep -a "per-point\|maxiter 10\|legacy_exact" | cut -c1-260; python3 tools/legacy_probe.py 1e-4 0 2>&1 | grep -a "per-point\|maxiter 10\|legacy_exact" | cut -c1-260; } > profiles/${TAG}_legacy_probe.txt
bash tools/pipe_columns.sh 4096 > profiles/${TAG}_pipe8_n4096_columns_pairs.txt 2>&1
bash tools/c5_timeline.sh gpurun_out/c5_timeline_$TAG > $O/c5_timeline.log 2>&1; cp gpurun_out/c5_timeline_$TAG/timeline.txt profiles/${TAG}_c5_timeline_after.txt
python3 tools/gallery_first_call.py > $O/gallery_first_call.txt 2>&1
{ [ -x tools/launch_floor ] && tools/launch_floor; } > profiles/${TAG}_direct_batch_floor.txt 2> $O/floor.err
timeout 600 python3 tools/fuzz_nlml.py 60 7 > $O/fuzz_nlml.txt 2>&1; tail -4 $O/fuzz_nlml.txt > profiles/${TAG}_fuzz_nlml_summary.txt
timeout 900 python3 tools/fuzz_gallery.py 400 7 > $O/fuzz_gallery.txt 2>&1; { grep -c " ok:" $O/fuzz_gallery.txt; grep "FAIL" $O/fuzz_gallery.txt | head; tail -1 $O/fuzz_gallery.txt; } > profiles/${TAG}_fuzz_gallery_summary.txt
for n in 1024 2048 4096; do bash tools/fit_trace.sh $n > profiles/${TAG}_fit_trace_$n.txt 2>&1; done
FUZZ_DIRECT_CASES=12 timeout 1500 python3 tools/fuzz_gpu.py 200 3 > $O/fuzz.txt 2>&1; { grep -c "rel err" $O/fuzz.txt; grep "FAIL" $O/fuzz.txt | head; grep "worst" $O/fuzz.txt; sort -t'e' -k1 $O/fuzz.txt | grep "rel err" | awk '{print}' | sort -k14 -g | tail -5; tail -4 $O/fuzz.txt; } > profiles/${TAG}_fuzz_summary.txt
tail -3 profiles/${TAG}_latencies.txt; cat profiles/${TAG}_configs_c2_c5.jsonl | cut -c1-300
# only gpurun_out/ travels back from the GPU box: take a copy of everything this script put under profiles/
mkdir -p $O/profiles; cp profiles/${TAG}_* $O/profiles/
