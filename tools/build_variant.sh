#!/bin/bash
# Build a variant of libgcpx.so with ONE source compiled under extra flags (ablations / tuning): tools/build_variant.sh <name> <source> <flags...>
#   -> video-gcp_amd/libgcpx_<name>.so (load it with GCPX_LIB=...); the other objects come from the regular build (csrc/build/)
set -e
cd "$(dirname "$0")/../video-gcp_amd/csrc"
name=$1; src=$2; shift 2
. ./sources.sh
extra="$(gcpx_flags_for $src)"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-variable $extra "$@" -c $src.hip -o build/variant_${src}_$name.o
objs=""
for f in $GCPX_SOURCES; do
  if [ $f = $src ]; then objs="$objs build/variant_${src}_$name.o"; else objs="$objs build/$f.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC $objs -ldl -o ../libgcpx_$name.so
echo "built video-gcp_amd/libgcpx_$name.so"
