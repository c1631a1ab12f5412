#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p $out
cd $root
python scripts/fetch_calibration.py r03 > $out/r03_calib2.log 2>&1
tail -30 $out/r03_calib2.log
V=fujiyama-renderer_amd/lib_var
python scripts/exp.py r03_exp5 \
  'dragon||--steps 5 --warmup 2' \
  "furry_phase|FJGPU_LIBDIR=$V/phase FJGPU_PHASE_STATS=1|--workload furry --steps 1 --warmup 0 --no-pmc" \
  'furry||--workload furry --steps 2 --warmup 1'
