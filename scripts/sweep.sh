#!/bin/bash
# usage: scripts/sweep.sh VAR v1 v2 ...   (other FJGPU_* taken from the environment)
var=$1; shift
for v in "$@"; do
  export $var=$v
  python bench.py --steps 2 --warmup 1 --cpu-tiles 0 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$var=$v', round(d['value'],1), round(d['ms_per_step'],1), d['config']['ms_last_frame_rank0']['trace'])"
done
