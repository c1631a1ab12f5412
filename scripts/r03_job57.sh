#!/bin/bash
# last run of the round: smoke(), the headline bench line exactly as the driver runs it (+ per-rank costs), kernel statistics + PMC of C3
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out
python __graft_entry__.py smoke > $out/r03_smoke.txt 2>&1; tail -n 2 $out/r03_smoke.txt
python bench.py --steps 20 --warmup 5 --rank-costs 8 > $out/r03_bench_dragon1080p.json 2> $out/r03_bench_dragon.err
tail -c 600 $out/r03_bench_dragon1080p.json
bash scripts/prof.sh r03_bench_dragon1080p > $out/r03_prof_dragon.log 2>&1
