#!/bin/bash
# Build ab_libs/<name>.so = the library with ONE source recompiled under extra flags (A/B runs: tools/ab.sh, tools/ab_case.sh).
#   tools/mk_variant.sh <name> <source.hip> [flags...]
set -eu
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
mkdir -p ab_libs /tmp/abobj
o=/tmp/abobj/$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c dagl_amd/csrc/$src -o $o
objs=$(ls dagl_amd/csrc/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $o -o ab_libs/$name.so
echo ab_libs/$name.so
