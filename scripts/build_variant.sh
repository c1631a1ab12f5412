#!/bin/bash
# usage: scripts/build_variant.sh NAME [-DFOO ...]   -> fujiyama-renderer_amd/lib/var/NAME/libfjgpu.so
# (perf experiments: the kernels translation unit rebuilt with extra macros, the rest reused)
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/fujiyama-renderer_amd/csrc
out=$root/fujiyama-renderer_amd/lib/var/$name
mkdir -p $out /tmp/var_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I$root/include -I$src/device "$@" -c -o /tmp/var_$name/k.o $src/device/fjgpu_kernels.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libfjgpu.so /tmp/var_$name/k.o $src/build/fjgpu_api.hip.o $src/build/fjgpu_build.o $src/build/fjgpu_xform.o $src/build/fjgpu_curve_build.o -pthread
echo built $out/libfjgpu.so
