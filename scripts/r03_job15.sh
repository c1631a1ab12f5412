cd ${GRAFT_REPO_ROOT:-/root/repo}
python -m pytest tests -m gpu -x -q > gpurun_out/r03_pytest_gpu_15.txt 2>&1
tail -3 gpurun_out/r03_pytest_gpu_15.txt
python scripts/exp.py r03_exp15 'dragon||--steps 10 --warmup 3' 'buddhas||--workload buddhas --steps 5 --warmup 2 --no-pmc' 'ibl||--workload ibl --steps 3 --warmup 1 --no-pmc' 'teapot||--workload teapot --steps 5 --warmup 2 --no-pmc'
