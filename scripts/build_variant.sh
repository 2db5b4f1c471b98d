#!/bin/bash
# A second build of the library with extra compiler definitions, for A/B runs on one box (OPAL_LIB_PATH=build/<name>/libopal.so):
#   scripts/build_variant.sh <name> [-DOPAL_...=... ...]
# Sources are copied into build/<name>/ (git-ignored; it travels to the GPU box with the snapshot) and built there.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
D=$ROOT/build/$NAME
mkdir -p $D
cp $ROOT/opal_amd/csrc/*.cpp $ROOT/opal_amd/csrc/*.hpp $ROOT/opal_amd/csrc/*.hip $ROOT/opal_amd/csrc/*.inc $ROOT/opal_amd/csrc/Makefile $ROOT/opal_amd/csrc/libopal.map $D/
make -s -j8 -C $D libopal.so HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include $*"
ls -la $D/libopal.so
