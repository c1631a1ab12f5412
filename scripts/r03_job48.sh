#!/bin/bash
# final artifacts of round 3 after the curve-walk changes (quantised nodes + instance level in LDS for the curve instantiations):
# the GPU suite, C5's kernel statistics / PMC / bench line, every workload, C5's whole-frame parity
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out
timeout 1700 python -m pytest tests -m gpu -q > $out/r03_pytest_verbose.log 2>&1
tail -n 3 $out/r03_pytest_verbose.log > $out/r03_pytest_gpu.txt; cat $out/r03_pytest_gpu.txt | cut -c1-300
bash scripts/prof.sh r03_furry1080p --workload furry > $out/r03_prof_furry.log 2>&1
python bench.py --workload furry --steps 3 --warmup 1 > $out/r03_bench_furry1080p.json 2> $out/r03_bench_furry.err
bash scripts/all_workloads.sh > $out/r03_all_workloads.txt 2>&1
cat $out/r03_all_workloads.txt
echo "== furry" > $out/r03_full_frame_parity_furry.txt
timeout 1500 python scripts/full_frame_parity.py furry >> $out/r03_full_frame_parity_furry.txt 2>&1
tail -n 8 $out/r03_full_frame_parity_furry.txt
