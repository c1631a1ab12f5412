#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 1700 python -m pytest tests -m gpu -q > gpurun_out/r03_pytest_verbose.log 2>&1
tail -n 8 gpurun_out/r03_pytest_verbose.log | cut -c1-300
tail -n 3 gpurun_out/r03_pytest_verbose.log > gpurun_out/r03_pytest_gpu.txt
