#!/bin/bash
# usage: scripts/build_variant.sh NAME [-DFOO ...]   -> fujiyama-renderer_amd/lib/var/NAME/libfjgpu.so
# (perf experiments: the device library rebuilt with extra macros)
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/fujiyama-renderer_amd/csrc
out=$root/fujiyama-renderer_amd/lib/var/$name
o=/tmp/var_$name
rm -rf $o $out; mkdir -p $out $o
for f in fjgpu_kernels fjgpu_api fjgpu_lbvh; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I$root/include -I$src/device "$@" -c -o $o/$f.o $src/device/$f.hip &
done
for f in fjgpu_build fjgpu_xform fjgpu_curve_build; do
  g++ -O3 -std=c++17 -ffp-contract=off -fPIC -I$root/include -I$src/device "$@" -c -o $o/$f.o $src/device/$f.cc &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libfjgpu.so $o/*.o -pthread
echo built $out/libfjgpu.so
