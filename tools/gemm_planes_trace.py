"""Per-stage timeline of one workgroup of the planes GEMM (tuning aid; needs the trace build:
tools/build_variant.sh trace gemm_planes -DGP_ABLATE=1 -DGP_TRACE): python tools/gemm_planes_trace.py [M]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_gcp_amd import runtime as rt, packing as pk
lib = rt.load_library(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "video-gcp_amd", "libgcpx_trace.so"))
dev = torch.device("cuda")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
N, K = 2048, 1024
x = torch.randn(M, K, device=dev)
w = torch.randn(N, K) / K ** 0.5
wp = pk.pack_gemm(w).to(dev)
ws, e = pk.pack_gemm_split(w)
ws = ws.to(dev)
b = torch.zeros(N, device=dev)
H = N // 4
c, ho, co = (torch.zeros(M, H, device=dev) for _ in range(3))
trace = torch.zeros(8 * (K // 32) * 2, dtype=torch.int64, device=dev)
a = rt.GemmArgs()
s = a.src[0]; s.ptr, s.sb, s.sr, s.width = x.data_ptr(), 0, K, K
a.nsrc, a.M, a.N, a.K, a.rpb = 1, M, N, K, M
a.wpk, a.bias = wp.data_ptr(), b.data_ptr()
a.epi = rt.EPI_LSTM
a.c_prev, a.c_prev_stride, a.h_out, a.c_out, a.hb, a.hrow, a.h_copy = c.data_ptr(), H, ho.data_ptr(), co.data_ptr(), 0, H, trace.data_ptr()
a.wpk_split, a.w_split_log2 = ws.data_ptr(), e
nbytes, nexp = C.c_int64(), C.c_int64()
rt.check(lib.gcpx_gemm_planes_workspace(M, K, 1, C.byref(nbytes), C.byref(nexp)), "ws")
planes = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
exps = torch.empty(nexp.value, dtype=torch.int32, device=dev)
a.x_planes, a.x_exp, a.x_planes_bytes = planes.data_ptr(), exps.data_ptr(), nbytes.value
for _ in range(3):
    rt.check(lib.gcpx_gemm(C.byref(a), torch.cuda.current_stream().cuda_stream), "gemm")
torch.cuda.synchronize()
t = trace.cpu().view(8, -1, 2)
nk = int((t[0, :, 0] != 0).sum())
t = t[:, :nk]
t0 = int(t[:, 0, 0].min())
print(f"{nk} stages; per wavefront: [after the barrier -> after the last MFMA issue] and the gap to the next stage's start (shader cycles)")
for wv in range(8):
    comp = (t[wv, :, 1] - t[wv, :, 0]).float()
    gap = (t[wv, 1:, 0] - t[wv, :-1, 1]).float()
    print(f"wave {wv}: compute mean {comp.mean():7.0f} min {comp.min():6.0f} max {comp.max():6.0f}   gap mean {gap.mean():7.0f} min {gap.min():6.0f} max {gap.max():6.0f}   "
          f"total {int(t[wv, -1, 1] - t[wv, 0, 0])}")
print("stage starts of wave 0 / wave 4 (first 6):", [int(v) - t0 for v in t[0, :6, 0]], [int(v) - t0 for v in t[4, :6, 0]])
