#!/bin/bash
# usage (GPU box): scripts/round_artifacts.sh r03   -> gpurun_out/<tag>_* (copy what is judged into profiles/)
# kernel statistics + PMC of the headline bench (dragon) and of the other BASELINE configurations, their bench lines,
# every workload at full size, the N-rank code path on one GPU, and the whole-frame device-vs-oracle comparison.
tag=${1:-r04}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
cd $root
bash scripts/prof.sh ${tag}_bench_dragon1080p > $out/${tag}_prof_dragon.log 2>&1
bash scripts/prof.sh ${tag}_furry1080p --workload furry > $out/${tag}_prof_furry.log 2>&1
bash scripts/prof.sh ${tag}_cornell1080p --workload cornell > $out/${tag}_prof_cornell.log 2>&1
bash scripts/prof.sh ${tag}_buddhas720p --workload buddhas > $out/${tag}_prof_buddhas.log 2>&1
python bench.py --steps 20 --warmup 5 --rank-costs 8 > $out/${tag}_bench_dragon1080p.json 2> $out/${tag}_bench_dragon.err
python bench.py --workload furry --steps 3 --warmup 1 > $out/${tag}_bench_furry1080p.json 2> $out/${tag}_bench_furry.err
python bench.py --workload cornell --steps 3 --warmup 1 > $out/${tag}_bench_cornell1080p.json 2> $out/${tag}_bench_cornell.err
python bench.py --workload buddhas --steps 5 --warmup 2 > $out/${tag}_bench_buddhas720p.json 2> $out/${tag}_bench_buddhas.err
python bench.py --dry-ranks 8 --steps 1 --warmup 1 > $out/${tag}_bench_dry_ranks8.json 2> $out/${tag}_bench_dry.err
bash scripts/all_workloads.sh > $out/${tag}_all_workloads.txt 2>&1
for w in buddhas dragon cornell furry ibl; do
  echo "== $w" >> $out/${tag}_full_frame_parity.txt
  timeout 1500 python scripts/full_frame_parity.py $w >> $out/${tag}_full_frame_parity.txt 2>&1
done
tail -n 40 $out/${tag}_full_frame_parity.txt
cat $out/${tag}_all_workloads.txt
