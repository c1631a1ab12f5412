#!/bin/bash
# usage: scripts/build_variant.sh NAME "-DFLAG ..." ["hipcc-only flags"]   -> fujiyama-renderer_amd/lib_var/NAME/{libfjgpu.so,libfjscene.so}
# An experiment build of the product libraries next to the default one; select it at run time
# with FJGPU_LIBDIR=fujiyama-renderer_amd/lib_var/NAME (ffi.py).  Built artefacts travel with gpurun.
set -e
name=$1; flags=$2; hipflags=$3
root=$(cd "$(dirname "$0")/.." && pwd)
pkg=$root/fujiyama-renderer_amd
make -s -C $pkg/csrc -j8 EXTRA="$flags" HIPEXTRA="$hipflags" OBJ=$pkg/csrc/build/var_$name LIBDIR=$pkg/lib_var/$name BINDIR=$pkg/csrc/build/var_$name/bin all
echo "built $pkg/lib_var/$name"
