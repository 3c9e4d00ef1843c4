# Sources of libgcpx.so and their per-file flags — the ONE list csrc/build.sh, tools/build_variant.sh and tools/isa_hazard_scan.py read.
GCPX_SOURCES="conv3x3 conv3x3_split conv3x3_head_split conv3x3_head32 conv_enc conv_enc_split gemm gemm_split gemm_planes mlp mlp_bwd misc loss wgrad wgrad_conv wgrad_conv_split wgrad_rows_split wgrad_image split_pack backward scalar_f32 adaptive aux metrics comm"
# built without SLP vectorisation = without packed-f32 VALU instructions (profiles/r05_head_store_hazard.txt, scalar_f32.hip)
GCPX_NO_SLP="conv3x3_head_split conv3x3_head32 loss scalar_f32"
GCPX_HEADERS="common.h gemm_tile.h split_tr.h split_mfma.h split_common.h ../../include/gcpx.h"
gcpx_flags_for() {
  for s in $GCPX_NO_SLP; do if [ "$s" = "$1" ]; then echo "-fno-slp-vectorize"; return; fi; done
  echo ""
}
