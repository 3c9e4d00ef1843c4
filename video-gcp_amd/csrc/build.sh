#!/bin/bash
# Build libgcpx.so (gfx950 only) in-tree: video-gcp_amd/libgcpx.so
set -e
cd "$(dirname "$0")"
OUT=../libgcpx.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-variable ${GCPX_EXTRA_FLAGS}"
mkdir -p build
pids=()
for f in conv3x3 conv3x3_split conv3x3_head_split conv_enc conv_enc_split gemm gemm_split gemm_planes mlp mlp_bwd misc loss wgrad wgrad_conv wgrad_conv_split wgrad_rows_split wgrad_image split_pack backward adaptive aux metrics comm; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ common.h -nt build/$f.o ] || [ gemm_tile.h -nt build/$f.o ] || [ split_tr.h -nt build/$f.o ] || [ split_mfma.h -nt build/$f.o ] || [ ../../include/gcpx.h -nt build/$f.o ]; then
    # (the output head and the likelihood kernels carry no packed-f32 VALU instructions: conv3x3_head_split.hip)
    extra=""; if [ $f = conv3x3_head_split ] || [ $f = loss ]; then extra="-fno-slp-vectorize"; fi
    hipcc $FLAGS $extra -c $f.hip -o build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC build/conv3x3.o build/conv3x3_split.o build/conv3x3_head_split.o build/conv_enc.o build/conv_enc_split.o build/gemm.o build/gemm_split.o build/gemm_planes.o build/mlp.o build/mlp_bwd.o build/misc.o build/loss.o build/wgrad.o build/wgrad_conv.o build/wgrad_conv_split.o build/wgrad_rows_split.o build/wgrad_image.o build/split_pack.o build/backward.o build/adaptive.o build/aux.o build/metrics.o build/comm.o -ldl -o $OUT
echo "built $(realpath $OUT)"
