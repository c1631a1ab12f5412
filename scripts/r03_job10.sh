#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p $out
cd $root
python -m pytest tests -m gpu -x -q > $out/r03_pytest_gpu_10.txt 2>&1
tail -3 $out/r03_pytest_gpu_10.txt
python bench.py --steps 10 --warmup 3 --rank-costs 8 > $out/r03_bench_dragon.json 2> $out/r03_bench_dragon.err
python scripts/exp.py r03_exp10 \
  'cornell||--workload cornell --steps 2 --warmup 1 --no-pmc' \
  'rank0of8||--as-rank-of 8 --steps 5 --warmup 2 --no-pmc' \
  'rank0of8_grab128|FJGPU_TRAV_GRAB=128|--as-rank-of 8 --steps 5 --warmup 2 --no-pmc' \
  'rank0of8_grab64|FJGPU_TRAV_GRAB=64|--as-rank-of 8 --steps 5 --warmup 2 --no-pmc' \
  'rank0of8_blocks4|FJGPU_ANYHIT_BLOCKS=4|--as-rank-of 8 --steps 5 --warmup 2 --no-pmc'
python - <<PY
import json
d=json.load(open("$out/r03_bench_dragon.json"))
print(d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["binding_resource"], d["config"].get("rank_costs_ms"))
PY
