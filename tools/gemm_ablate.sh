#!/bin/bash
# per-kernel times of the planes GEMM and its ablation builds (tools/build_variant.sh noload / nomfma gemm_planes -DGP_ABLATE=1 / 2):
#   tools/gemm_ablate.sh [rows ...]
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for v in "" _noload _nomfma; do
  [ -f $R/video-gcp_amd/libgcpx$v.so ] || continue
  rm -rf /tmp/prof$v
  GCPX_LIB=$R/video-gcp_amd/libgcpx$v.so rocprofv3 --kernel-trace -d /tmp/prof$v -o t --output-format csv -- python3 $R/tools/time_gemm_split.py "$@" > /tmp/out$v.txt 2>&1
  echo "== variant '$v'"; grep "planes\|split " /tmp/out$v.txt
  python3 $R/tools/kstats.py $(find /tmp/prof$v -name "*kernel_trace.csv" | head -1) planes split_rows
done
