#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
A='--no-pmc --steps 3 --warmup 1 --workload furry'
timeout 1500 python scripts/exp.py r03_exp49 \
  "steps4|FJGPU_TRAV_STEPS_CURVES=4|$A" \
  "steps8|FJGPU_TRAV_STEPS_CURVES=8|$A" \
  "steps12|FJGPU_TRAV_STEPS_CURVES=12|$A" \
  "refill8|FJGPU_TRAV_REFILL_CURVES=8|$A" \
  "refill24|FJGPU_TRAV_REFILL_CURVES=24|$A" \
  "lw32|FJGPU_TRAV_LEAFWAIT=32|$A" \
  "lw48|FJGPU_TRAV_LEAFWAIT=48|$A"
