#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out
timeout 1700 python -m pytest tests -m gpu -q > $out/r03_pytest_verbose.log 2>&1
tail -n 3 $out/r03_pytest_verbose.log > $out/r03_pytest_gpu.txt; cat $out/r03_pytest_gpu.txt | cut -c1-300
python bench.py --workload furry --steps 3 --warmup 1 > $out/r03_bench_furry1080p.json 2> $out/r03_bench_furry.err
tail -c 300 $out/r03_bench_furry1080p.json
