#!/bin/bash
# round 3, GPU job 1: GPU suite, FETCH_SIZE calibration, bound experiments on the any-hit walk (occupancy sweep,
# ALU padding, phase statistics), bench line with the reworked roofline object
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p $out
cd $root
python -m pytest tests -m gpu -x -q > $out/r03_pytest_gpu_1.txt 2>&1
tail -3 $out/r03_pytest_gpu_1.txt
python scripts/fetch_calibration.py r03 > $out/r03_calib.log 2>&1
tail -5 $out/r03_calib.log
V=fujiyama-renderer_amd/lib_var
python scripts/exp.py r03_exp1 \
  'base||--steps 5 --warmup 2' \
  'blocks4|FJGPU_ANYHIT_BLOCKS=4|--steps 5 --warmup 2' \
  'blocks3|FJGPU_ANYHIT_BLOCKS=3|--steps 5 --warmup 2' \
  "pad16|FJGPU_LIBDIR=$V/pad16|--steps 5 --warmup 2" \
  "pad32|FJGPU_LIBDIR=$V/pad32|--steps 5 --warmup 2" \
  "phase|FJGPU_LIBDIR=$V/phase FJGPU_PHASE_STATS=1|--steps 1 --warmup 0 --no-pmc"
python bench.py --steps 10 --warmup 3 > $out/r03_bench1.json 2> $out/r03_bench1.err
tail -c 3000 $out/r03_bench1.json
