#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python scripts/exp.py r03_exp42 \
  'furry_base||--no-pmc --steps 3 --warmup 1 --workload furry' \
  'furry_swap2|FJGPU_EXP_SWAP2=1|--no-pmc --steps 3 --warmup 1 --workload furry'
