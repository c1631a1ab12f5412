#!/bin/bash
# every workload at its default (full) size, one frame each: catches scale-only failures
for w in teapot buddhas dragon furry ibl cornell motion arealights; do
  extra="--no-pmc"
  timeout 600 python bench.py --workload $w --steps 2 --warmup 1 --cpu-tiles 0 $extra 2>gpurun_out/all_$w.err | tail -1 | python -c "
import json,sys
try:
    d=json.load(sys.stdin); print('$w', round(d['value'],1), 'Mray/s', round(d['ms_per_step'],1), 'ms', d['config']['rays_per_frame_rank0'], 'prep', round(d['config']['prepare_seconds'],2))
except Exception as e:
    print('$w FAILED', e)
"
done
