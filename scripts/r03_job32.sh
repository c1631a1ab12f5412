#!/bin/bash
# instance level in LDS for the persistent closest-hit / general shadow walks of mesh scenes (template choice) against global memory
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 1500 python -m pytest tests -m gpu -x -v > gpurun_out/r03_pytest_verbose.log 2>&1
tail -n 30 gpurun_out/r03_pytest_verbose.log | cut -c1-200
A="--steps 3 --warmup 1 --no-pmc"
python scripts/exp.py r03_exp32 \
  "dragon_lds||$A" \
  "dragon_global|FJGPU_NO_INST_LDS=1|$A" \
  "buddhas_lds||--workload buddhas $A" \
  "ibl_lds||--workload ibl $A" \
  "ibl_global|FJGPU_NO_INST_LDS=1|--workload ibl $A" \
  "furry_lds||--workload furry --steps 2 --warmup 1 --no-pmc" \
  "motion_lds||--workload motion $A" \
  "arealights_lds||--workload arealights $A" \
  "arealights_global|FJGPU_NO_INST_LDS=1|--workload arealights $A"
