#!/bin/bash
# on the GPU box: bench every lib/var/*/libfjgpu.so (swapped into lib/), print one line each
root=$(cd "$(dirname "$0")/.." && pwd)
lib=$root/fujiyama-renderer_amd/lib
cp $lib/libfjgpu.so /tmp/libfjgpu_orig.so
for v in "$@"; do
  cp $lib/var/$v/libfjgpu.so $lib/libfjgpu.so
  python $root/bench.py --steps 2 --warmup 1 --cpu-tiles 0 $BENCH_ARGS 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$v', round(d['value'],1), round(d['ms_per_step'],1), d['config']['ms_last_frame_rank0'])"
done
cp /tmp/libfjgpu_orig.so $lib/libfjgpu.so
