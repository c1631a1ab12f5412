#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
cd $root
V=fujiyama-renderer_amd/lib_var
python scripts/exp.py r03_exp7 \
  'w4s8||--workload furry --steps 2 --warmup 1 --no-pmc' \
  "w2s8|FJGPU_LIBDIR=$V/w2s8|--workload furry --steps 2 --warmup 1 --no-pmc" \
  "w8s16|FJGPU_LIBDIR=$V/w8s16|--workload furry --steps 2 --warmup 1 --no-pmc" \
  "w4s24|FJGPU_LIBDIR=$V/w4s24|--workload furry --steps 2 --warmup 1 --no-pmc"
