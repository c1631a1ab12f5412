#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests -m gpu -q -x -k "curve or hair or frames_match or adaptive or motion" > gpurun_out/r03_job50_pytest.log 2>&1
tail -n 3 gpurun_out/r03_job50_pytest.log | cut -c1-300
A='--no-pmc --steps 3 --warmup 1 --workload furry'
timeout 1500 python scripts/exp.py r03_exp50 \
  "postpone||$A" \
  "postpone_steps4|FJGPU_TRAV_STEPS_CURVES=4|$A" \
  "postpone_steps8|FJGPU_TRAV_STEPS_CURVES=8|$A" \
  "postpone_refill8|FJGPU_TRAV_REFILL_CURVES=8|$A" \
  "postpone_lw32|FJGPU_TRAV_LEAFWAIT=32|$A" \
  "postpone_lw48|FJGPU_TRAV_LEAFWAIT=48|$A"
