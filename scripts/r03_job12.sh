#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
cd $root
python -m pytest tests -m gpu -x -q -k "face_normals or full_size or whole_frame" > $out/r03_pytest_gpu_12.txt 2>&1
tail -3 $out/r03_pytest_gpu_12.txt
python scripts/exp.py r03_exp12 \
  'facen||--steps 5 --warmup 2 --no-pmc' \
  'nofacen|FJGPU_NO_FACE_N=1|--steps 5 --warmup 2 --no-pmc' \
  'buddhas||--workload buddhas --steps 5 --warmup 2 --no-pmc' \
  'cornell||--workload cornell --steps 2 --warmup 1 --no-pmc'
