#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
python scripts/exp.py r03_exp11 \
  'sort7||--workload cornell --steps 2 --warmup 1 --no-pmc' \
  'sort5|FJGPU_RAY_SORT=5|--workload cornell --steps 2 --warmup 1 --no-pmc' \
  'sort4|FJGPU_RAY_SORT=4|--workload cornell --steps 2 --warmup 1 --no-pmc' \
  'sort3|FJGPU_RAY_SORT=3|--workload cornell --steps 2 --warmup 1 --no-pmc'
for f in sort7 sort5 sort4 sort3; do python -c "
import json; d=json.load(open('gpurun_out/r03_exp11.$f.json')); print('$f', d['ms_per_step'], d['config']['ms_last_frame_rank0']['ray_sort'])"; done
