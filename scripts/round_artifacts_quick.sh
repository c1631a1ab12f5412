#!/bin/bash
# usage (GPU box): scripts/round_artifacts_quick.sh r04b  -> the headline's kernel statistics + PMC + bench line with rank costs, C5's bench line and statistics
tag=${1:-r04}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
cd $root
bash scripts/prof.sh ${tag}_bench_dragon1080p > $out/${tag}_prof_dragon.log 2>&1
python bench.py --steps 20 --warmup 5 --rank-costs 8 > $out/${tag}_bench_dragon1080p.json 2> $out/${tag}_bench_dragon.err
bash scripts/prof.sh ${tag}_furry1080p --workload furry > $out/${tag}_prof_furry.log 2>&1
python bench.py --workload furry --steps 3 --warmup 1 > $out/${tag}_bench_furry1080p.json 2> $out/${tag}_bench_furry.err
tail -c 600 $out/${tag}_bench_dragon1080p.json
