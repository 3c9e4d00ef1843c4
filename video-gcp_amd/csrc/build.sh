#!/bin/bash
# Build libgcpx.so (gfx950 only) in-tree: video-gcp_amd/libgcpx.so.  Incremental: an object is rebuilt when its source, a shared header,
# this script / sources.sh, or the flags it was compiled with (stamped beside it) changed.  GCPX_REBUILD=1 rebuilds everything.
set -e
cd "$(dirname "$0")"
. ./sources.sh
OUT=../libgcpx.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-variable ${GCPX_EXTRA_FLAGS}"
mkdir -p build
pids=()
for f in $GCPX_SOURCES; do
  want="$FLAGS $(gcpx_flags_for $f)"
  stale=0
  if [ -n "$GCPX_REBUILD" ] || [ ! -f build/$f.o ] || [ ! -f build/$f.flags ] || [ "$(cat build/$f.flags)" != "$want" ]; then stale=1; fi
  for dep in $f.hip $GCPX_HEADERS; do
    if [ -f $dep ] && [ $dep -nt build/$f.o ]; then stale=1; fi
  done
  if [ $stale = 1 ]; then
    ( hipcc $want -c $f.hip -o build/$f.o && echo "$want" > build/$f.flags ) &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
objs=""
for f in $GCPX_SOURCES; do objs="$objs build/$f.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC $objs -ldl -o $OUT
echo "built $(realpath $OUT)"
