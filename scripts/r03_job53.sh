#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests -m gpu -q -x -k "split or crowd or frames_match or instance_level or area or edge or groups or whole_frame_c2" > gpurun_out/r03_job53_pytest.log 2>&1
tail -n 3 gpurun_out/r03_job53_pytest.log | cut -c1-300
timeout 1500 python scripts/exp.py r03_exp53 \
  "buddhas_f32twin||--no-pmc --steps 8 --warmup 3 --workload buddhas" \
  "buddhas_nosplit|FJGPU_SPLIT_SHADOW=0|--no-pmc --steps 8 --warmup 3 --workload buddhas" \
  "dragon||--no-pmc --steps 8 --warmup 3" \
  "crowd||--no-pmc --steps 8 --warmup 3 --workload crowd"
