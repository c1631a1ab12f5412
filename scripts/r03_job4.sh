#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p $out
cd $root
V=fujiyama-renderer_amd/lib_var
python scripts/exp.py r03_exp4 \
  'lean4||--steps 5 --warmup 2 --no-pmc' \
  'shadefull|FJGPU_SHADE_FULL=1|--steps 5 --warmup 2 --no-pmc' \
  "lean3|FJGPU_LIBDIR=$V/shade3|--steps 5 --warmup 2 --no-pmc" \
  'phased|FJGPU_PHASED_CLOSEST=1 FJGPU_RAY_SORT=0|--steps 5 --warmup 2 --no-pmc' \
  "closest4|FJGPU_LIBDIR=$V/closest4|--steps 5 --warmup 2 --no-pmc" \
  'ibl_lean4||--workload ibl --steps 3 --warmup 1 --no-pmc'
python -m pytest tests -m gpu -x -q > $out/r03_pytest_gpu_4.txt 2>&1
tail -3 $out/r03_pytest_gpu_4.txt
