"""The few-row LSTM-layer GEMM of the tree levels (N = 2048, K = 1024, EPI_LSTM; gemm_kernel<1,1,true,true>) at M rows, with the
weights either the same 8.4 MB every launch (hot: L2 / Infinity Cache) or cycling through 48 different sets (400 MB: cold, as the
forward sees them: 290 MB of tree weights per forward).  python tools/r06/gemm_rows.py <rows> <hot|cold>"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from video_gcp_amd import runtime as rt, packing as pk
lib = rt.load_library()
dev = torch.device("cuda")
M, mode = int(sys.argv[1]), sys.argv[2]
N, K = 2048, 1024
nset = 48 if mode == "cold" else 1
w = torch.randn(N, K) / K ** 0.5
wp0 = pk.pack_gemm(w).to(dev)
wps = [wp0.clone() for _ in range(nset)]
x = torch.randn(M, K, device=dev)
b = torch.randn(N, device=dev)
H = N // 4
c, ho, co = torch.randn(M, H, device=dev), torch.zeros(M, H, device=dev), torch.zeros(M, H, device=dev)
args = []
for wp in wps:
    a = rt.GemmArgs()
    s = a.src[0]
    s.ptr, s.sb, s.sr, s.width = x.data_ptr(), 0, K, K
    a.nsrc, a.M, a.N, a.K, a.rpb = 1, M, N, K, M
    a.wpk, a.bias, a.out, a.ob, a.orow, a.epi = wp.data_ptr(), b.data_ptr(), None, 0, N, rt.EPI_LSTM
    a.c_prev, a.c_prev_stride, a.h_out, a.c_out, a.hb, a.hrow = c.data_ptr(), H, ho.data_ptr(), co.data_ptr(), 0, H
    args.append(a)
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for a in args:
        rt.check(lib.gcpx_gemm(C.byref(a), st.cuda_stream), "gemm")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 4
    e0.record(st)
    for _ in range(reps):
        for a in args:
            rt.check(lib.gcpx_gemm(C.byref(a), st.cuda_stream), "gemm")
    e1.record(st)
    st.synchronize()
    us = e0.elapsed_time(e1) / (reps * len(args)) * 1e3
print(f"M={M} {mode}: {us:.2f} us per launch back to back, {4 * N * K / us / 1e6:.2f} TB/s of weights")
