#!/bin/bash
# on the GPU box: run a command with lib/var/$1/libfjgpu.so swapped in
root=$(cd "$(dirname "$0")/.." && pwd)
lib=$root/fujiyama-renderer_amd/lib
v=$1; shift
cp $lib/libfjgpu.so /tmp/libfjgpu_orig.so
cp $lib/var/$v/libfjgpu.so $lib/libfjgpu.so
"$@"
cp /tmp/libfjgpu_orig.so $lib/libfjgpu.so
