"""Time gcpx_gemm on the shapes of the gcp_tree forward (c2): tuning aid."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_gcp_amd import runtime as rt, packing as pk
lib = rt.load_library()
dev = torch.device("cuda")
shapes = []
for l in range(7):
    M = 16 * 2 ** l
    shapes += [(f"L{l} lstm", M, 2048, 1024, 1, rt.EPI_LSTM), (f"L{l} merge x6", M, 512, 1024, 6, rt.EPI_NONE),
               (f"L{l} embed", M, 512, 768, 1, rt.EPI_NONE), (f"L{l} out", M, 128, 512, 1, rt.EPI_NONE)]
shapes += [("enc head traj", 1280, 128, 2048, 1, rt.EPI_NONE), ("dec input", 2032, 2048, 128, 1, rt.EPI_NONE),
           ("seq conv", 1280, 128, 384, 1, rt.EPI_NONE)]
st = torch.cuda.Stream()
tot = 0.0
with torch.cuda.stream(st):
    for name, M, N, K, nb, epi in shapes:
        x = torch.randn(M, K * nb, device=dev)
        w = torch.randn(nb, N, K, device=dev) / K ** 0.5
        wp = torch.stack([pk.pack_gemm(w[z]) for z in range(nb)]).contiguous()
        b = torch.randn(nb, N, device=dev)
        out = torch.zeros(M, N * nb, device=dev)
        a = rt.GemmArgs()
        s = a.src[0]
        s.ptr, s.sb, s.sr, s.width = x.data_ptr(), 0, K * nb, K
        a.nsrc, a.M, a.N, a.K, a.rpb = 1, M, N, K, M
        a.wpk, a.bias, a.out, a.ob, a.orow, a.epi = wp.data_ptr(), b.data_ptr(), out.data_ptr(), 0, N * nb, epi
        if nb > 1:
            a.nbatch, a.z_src_off, a.z_w_off, a.z_bias_off, a.z_out_off = nb, K, wp[0].numel(), N, N
        if epi == rt.EPI_LSTM:
            H = N // 4
            c, ho, co = torch.randn(M, H, device=dev), torch.zeros(M, H, device=dev), torch.zeros(M, H, device=dev)
            a.c_prev, a.c_prev_stride, a.h_out, a.c_out, a.hb, a.hrow = c.data_ptr(), H, ho.data_ptr(), co.data_ptr(), 0, H
        for _ in range(3):
            rt.check(lib.gcpx_gemm(C.byref(a), st.cuda_stream), name)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        n = 20
        for _ in range(n):
            rt.check(lib.gcpx_gemm(C.byref(a), st.cuda_stream), name)
        e1.record(st)
        st.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        fl = 2.0 * M * N * K * nb
        mult = 3 if "lstm" in name else 1
        tot += us * mult
        print(f"{name:16s} M={M:5d} N={N:5d} K={K:5d} nb={nb}: {us:8.1f} us  {fl / us / 1e6:7.1f} TF  W={4*N*K*nb/1e6:6.1f} MB -> {4*N*K*nb/us/1e6:6.2f} TB/s")
print("sum over forward (lstm x3):", tot, "us")
