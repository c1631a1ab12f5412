#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03_pytest_verbose.log 2>&1
tail -n 6 gpurun_out/r03_pytest_verbose.log | cut -c1-200
