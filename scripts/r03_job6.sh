#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p $out
cd $root
python -m pytest tests -m gpu -x -q -k "curve or furry or c5 or hair or adaptive" > $out/r03_pytest_gpu_6.txt 2>&1
tail -3 $out/r03_pytest_gpu_6.txt
V=fujiyama-renderer_amd/lib_var
python scripts/exp.py r03_exp6 \
  'coop||--workload furry --steps 2 --warmup 1 --no-pmc' \
  "nocoop|FJGPU_LIBDIR=$V/nocoop|--workload furry --steps 2 --warmup 1 --no-pmc" \
  "coop2w|FJGPU_LIBDIR=$V/coop2w|--workload furry --steps 2 --warmup 1 --no-pmc" \
  'coop_lw24|FJGPU_TRAV_LEAFWAIT=24|--workload furry --steps 2 --warmup 1 --no-pmc' \
  'coop_lw16|FJGPU_TRAV_LEAFWAIT=16|--workload furry --steps 2 --warmup 1 --no-pmc' \
  'coop_lw56|FJGPU_TRAV_LEAFWAIT=56|--workload furry --steps 2 --warmup 1 --no-pmc'
