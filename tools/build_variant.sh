#!/bin/bash
# Build an A/B variant of libonepose_hip.so:  tools/build_variant.sh <name> <git rev> <csrc file> [<csrc file> ...]
# = the working tree's csrc with the listed files taken from <git rev>; result: onepose_st_amd/lib/variants/libonepose_hip_<name>.so
# EXTRA="-D..." in the environment adds compiler flags; <git rev> may be "-" with no files (working tree + EXTRA only).
# (load it with OPHIP_LIB=<path>; `tools/box.sh <name> ab:R:S:variants` runs variants interleaved on one box).  Variants are scratch: git-ignored like every .so.
set -e
name=$1; rev=$2; shift 2
root=$(cd $(dirname $0)/.. && pwd)
tmp=$(mktemp -d)
mkdir -p $tmp/onepose_st_amd $root/onepose_st_amd/lib/variants
cp -r $root/onepose_st_amd/csrc $tmp/onepose_st_amd/csrc
cp -r $root/include $tmp/include
rm -rf $tmp/onepose_st_amd/csrc/build
[ "$rev" = "-" ] || for f in "$@"; do git -C $root show $rev:onepose_st_amd/csrc/$f > $tmp/onepose_st_amd/csrc/$f; done
make -C $tmp/onepose_st_amd/csrc -j8 EXTRA="$EXTRA" OUT=$root/onepose_st_amd/lib/variants/libonepose_hip_$name.so > $tmp/build.log 2>&1 || { tail -20 $tmp/build.log; exit 1; }
rm -rf $tmp
echo built $root/onepose_st_amd/lib/variants/libonepose_hip_$name.so
