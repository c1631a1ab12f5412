#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests -m gpu -q -x -k "curve or hair or frames_match or adaptive or instance_level" > gpurun_out/r03_job51_pytest.log 2>&1
tail -n 3 gpurun_out/r03_job51_pytest.log | cut -c1-300
A='--no-pmc --steps 3 --warmup 1 --workload furry'
timeout 1500 python scripts/exp.py r03_exp51 "anyonly||$A"
