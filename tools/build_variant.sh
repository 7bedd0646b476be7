#!/bin/bash
# Build a variant of libsls_hip.so WITHOUT touching the tree's objects: tools/build_variant.sh <name> [make variables...]
#   tools/build_variant.sh trace "FAST=--offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSLS_TILE_W=16 -DSLS_TILE_H=16 -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function -munsafe-fp-atomics -fno-slp-vectorize -DSLS_TRACE"
# -> ./gpurun_tmp_<name>.so (travels to the GPU box with the snapshot; git ignores it).  EXTRA="-DFOO" is appended to
# both flag sets through the COMMON variable when given as EXTRA=...
set -e
NAME=$1; shift
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
TMP=$(mktemp -d /tmp/sls_variant_XXXX)
mkdir -p $TMP/splat_loam_amd $TMP/include
cp -r $ROOT/splat_loam_amd/csrc $TMP/splat_loam_amd/csrc
cp $ROOT/include/*.h $TMP/include/
rm -f $TMP/splat_loam_amd/csrc/*.o
make -C $TMP/splat_loam_amd/csrc -j8 OUT=$ROOT/gpurun_tmp_$NAME.so "$@" > $TMP/build.log 2>&1 || { tail -20 $TMP/build.log; exit 1; }
rm -rf $TMP
ls -la $ROOT/gpurun_tmp_$NAME.so
