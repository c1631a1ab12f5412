#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
for cfg in "host_trap time 1" "host_trap time 1000" "host_trap time 10000" "stochastic cycles 1048576" "stochastic cycles 65536" "host_trap cycles 1048576" "stochastic time 1000"; do
  set -- $cfg
  timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $1 --pc-sampling-unit $2 --pc-sampling-interval $3 --kernel-trace --output-format csv -d $out/r03_pcsamp -- python $root/bench.py --workload teapot --steps 1 --warmup 0 --cpu-tiles 0 --no-pmc > /tmp/pcs.log 2>&1
  echo "$cfg rc=$? $(grep -i "not supported\|error" /tmp/pcs.log | head -2)"
  ls $out/r03_pcsamp/*/ 2>/dev/null | head -5
done
