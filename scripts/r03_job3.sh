#!/bin/bash
# round 3, GPU job 3: implicit camera rays, per-lane light lists, perm pad .51: GPU suite + A/B
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p $out
cd $root
python -m pytest tests -m gpu -x -q > $out/r03_pytest_gpu_3.txt 2>&1
tail -3 $out/r03_pytest_gpu_3.txt
V=fujiyama-renderer_amd/lib_var
python scripts/exp.py r03_exp3 \
  'all||--steps 5 --warmup 2 --no-pmc' \
  'nolists|FJGPU_LIGHT_LISTS=0|--steps 5 --warmup 2 --no-pmc' \
  'explicitcam|FJGPU_EXPLICIT_CAMERA_RAYS=1|--steps 5 --warmup 2 --no-pmc' \
  "noperm|FJGPU_LIBDIR=$V/noperm|--steps 5 --warmup 2 --no-pmc" \
  'buddhas||--workload buddhas --steps 5 --warmup 2 --no-pmc' \
  'buddhas_nolists|FJGPU_LIGHT_LISTS=0|--workload buddhas --steps 5 --warmup 2 --no-pmc' \
  'ibl||--workload ibl --steps 3 --warmup 1 --no-pmc' \
  'ibl_nolists|FJGPU_LIGHT_LISTS=0|--workload ibl --steps 3 --warmup 1 --no-pmc' \
  'furry||--workload furry --steps 2 --warmup 1 --no-pmc' \
  'cornell||--workload cornell --steps 2 --warmup 1 --no-pmc'
