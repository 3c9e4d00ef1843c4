for t in default 4,4,256 4,2,256 2,4,256 2,2,256 4,1,256 1,4,256; do
  echo "=== tile $t"
  if [ "$t" = default ]; then python tools/time_gemm.py 2>&1 | grep -E "^L[3456] (lstm|merge)|sum"; else GCPX_GEMM_TILE=$t python tools/time_gemm.py 2>&1 | grep -E "^L[456] (lstm|merge)"; fi
done
