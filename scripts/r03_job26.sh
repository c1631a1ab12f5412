#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
A="--steps 3 --warmup 1 --no-pmc"
python scripts/exp.py r03_exp26 \
  "cornell_lds2||--workload cornell $A" \
  "cornell_global2|FJGPU_NO_INST_LDS=1|--workload cornell $A" \
  "cornell_lds2_refill32|FJGPU_TRAV_REFILL=32|--workload cornell $A" \
  "cornell_lds2_refill48|FJGPU_TRAV_REFILL=48|--workload cornell $A"
FJGPU_LIBDIR=fujiyama-renderer_amd/lib_var/phase FJGPU_PHASE_STATS=1 python bench.py --workload cornell --steps 1 --warmup 0 --no-pmc \
  > gpurun_out/r03_exp26_phases.json 2> gpurun_out/r03_exp26_phases.err
grep "phased-walk" gpurun_out/r03_exp26_phases.err | tail -n 3
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -n 5
