#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
A='--no-pmc --steps 3 --warmup 1 --workload furry'
timeout 1500 python scripts/exp.py r03_exp52 "anyonly_nearest_first||$A"
