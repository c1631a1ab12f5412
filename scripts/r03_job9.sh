#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p $out
cd $root
python -m pytest tests -m gpu -x -q > $out/r03_pytest_gpu_9.txt 2>&1
tail -3 $out/r03_pytest_gpu_9.txt
python bench.py --steps 10 --warmup 3 --rank-costs 8 > $out/r03_bench_dragon.json 2> $out/r03_bench_dragon.err
python scripts/exp.py r03_exp9 \
  'cornell||--workload cornell --steps 2 --warmup 1' \
  'furry||--workload furry --steps 2 --warmup 1' \
  'buddhas||--workload buddhas --steps 5 --warmup 2' \
  'ibl||--workload ibl --steps 3 --warmup 1 --no-pmc'
python - <<PY
import json
d=json.load(open("$out/r03_bench_dragon.json"))
print(d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["binding_resource"], d["config"].get("rank_costs_ms"))
PY
