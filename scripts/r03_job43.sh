#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
A='--no-pmc --steps 3 --warmup 1 --workload furry'
timeout 1200 python scripts/exp.py r03_exp43 \
  "s20|FJGPU_LIBDIR=fujiyama-renderer_amd/lib_var/s20|$A" \
  "c0|FJGPU_LIBDIR=fujiyama-renderer_amd/lib_var/c0|$A" \
  "c0s40|FJGPU_LIBDIR=fujiyama-renderer_amd/lib_var/c0s40|$A"
