#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
A='--no-pmc --steps 3 --warmup 1 --workload furry'
timeout 1200 python scripts/exp.py r03_exp47 \
  "furry_ao1|FJGPU_LIBDIR=fujiyama-renderer_amd/lib_var/ao1|$A" \
  "furry_ao2|FJGPU_LIBDIR=fujiyama-renderer_amd/lib_var/ao2|$A" \
  "teapot_ao2|FJGPU_LIBDIR=fujiyama-renderer_amd/lib_var/ao2|--no-pmc --steps 5 --warmup 2 --workload teapot" \
  "teapot_base||--no-pmc --steps 5 --warmup 2 --workload teapot"
