#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests -m gpu -q -x -k "curve or hair or frames_match" > gpurun_out/r03_job45_pytest.log 2>&1
tail -n 3 gpurun_out/r03_job45_pytest.log | cut -c1-300
A='--no-pmc --steps 3 --warmup 1 --workload furry'
timeout 1200 python scripts/exp.py r03_exp45 \
  "furry_pend2||$A" \
  "furry_pend2_lw32|FJGPU_TRAV_LEAFWAIT=32|$A" \
  "furry_pend2_lw48|FJGPU_TRAV_LEAFWAIT=48|$A"
