#!/bin/bash
# round 3, GPU job 2: 8-wide any-hit walk + perm slabs: GPU suite, slab validation, A/B against the 4-wide walk, counter list
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p $out
cd $root
python -m pytest tests -m gpu -x -q > $out/r03_pytest_gpu_2.txt 2>&1
tail -3 $out/r03_pytest_gpu_2.txt
(cd /tmp && rocprofv3 -L 2>/dev/null | grep -i -E "TCC_EA0|TCC_BUBBLE|TCC_REQ|TCC_HIT|TCC_MISS|MALL|DRAM" | head -80) > $out/r03_tcc_counters.txt 2>&1
V=fujiyama-renderer_amd/lib_var
python scripts/exp.py r03_exp2 \
  'wide8||--steps 5 --warmup 2' \
  'wide4perm|FJGPU_NO_WIDE8=1|--steps 5 --warmup 2' \
  "validate4|FJGPU_LIBDIR=$V/validate FJGPU_NO_WIDE8=1 FJGPU_PHASE_STATS=1|--steps 1 --warmup 0 --no-pmc" \
  'wide8_blocks4|FJGPU_ANYHIT_BLOCKS=4|--steps 3 --warmup 1 --no-pmc' \
  'buddhas_wide8||--workload buddhas --steps 5 --warmup 2 --no-pmc' \
  'buddhas_wide4|FJGPU_NO_WIDE8=1|--workload buddhas --steps 5 --warmup 2 --no-pmc' \
  'ibl_wide8||--workload ibl --steps 3 --warmup 1 --no-pmc' \
  'ibl_wide4|FJGPU_NO_WIDE8=1|--workload ibl --steps 3 --warmup 1 --no-pmc'
