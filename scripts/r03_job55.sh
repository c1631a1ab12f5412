#!/bin/bash
# final artifacts after the join-slot change: GPU suite, C2's kernel statistics / PMC / bench line, C2 whole-frame parity, every workload
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out
timeout 1700 python -m pytest tests -m gpu -q > $out/r03_pytest_verbose.log 2>&1
tail -n 3 $out/r03_pytest_verbose.log > $out/r03_pytest_gpu.txt; cat $out/r03_pytest_gpu.txt | cut -c1-300
bash scripts/prof.sh r03_buddhas720p --workload buddhas > $out/r03_prof_buddhas.log 2>&1
python bench.py --workload buddhas --steps 5 --warmup 2 > $out/r03_bench_buddhas720p.json 2> $out/r03_bench_buddhas.err
echo "== buddhas" > $out/r03_full_frame_parity_buddhas.txt
timeout 900 python scripts/full_frame_parity.py buddhas >> $out/r03_full_frame_parity_buddhas.txt 2>&1
tail -n 5 $out/r03_full_frame_parity_buddhas.txt
bash scripts/all_workloads.sh > $out/r03_all_workloads.txt 2>&1
cat $out/r03_all_workloads.txt
