#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
V=fujiyama-renderer_amd/lib_var
A="--steps 5 --warmup 2 --no-pmc"
python scripts/exp.py r03_exp13 \
  "base||$A" \
  "pf1|FJGPU_LIBDIR=$V/pf1|$A" \
  "pf2|FJGPU_LIBDIR=$V/pf2|$A" \
  "pf3|FJGPU_LIBDIR=$V/pf3|$A" \
  "w5lds|FJGPU_LIBDIR=$V/w5lds|$A" \
  "w6|FJGPU_LIBDIR=$V/w6|$A" \
  "w6pf1|FJGPU_LIBDIR=$V/w6pf1|$A" \
  "w6pf3|FJGPU_LIBDIR=$V/w6pf3|$A" \
  "buddhas_w6|FJGPU_LIBDIR=$V/w6|--workload buddhas $A" \
  "buddhas_w6m5nosplit|FJGPU_LIBDIR=$V/w6m5 FJGPU_SPLIT_SHADOW=0|--workload buddhas $A" \
  "cornell_ph5|FJGPU_LIBDIR=$V/ph5|--workload cornell --steps 2 --warmup 1 --no-pmc" \
  "cornell_ph4lds|FJGPU_LIBDIR=$V/ph4lds|--workload cornell --steps 2 --warmup 1 --no-pmc" \
  "ibl_w6|FJGPU_LIBDIR=$V/w6|--workload ibl --steps 3 --warmup 1 --no-pmc"
