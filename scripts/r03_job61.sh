#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python bench.py --dry-ranks 4 --steps 1 --warmup 1 > gpurun_out/r03_bench_dry_ranks4.json 2> gpurun_out/r03_bench_dry.err
tail -c 700 gpurun_out/r03_bench_dry_ranks4.json; tail -n 3 gpurun_out/r03_bench_dry.err
