"""Time the row GEMM at the tree-level shapes, exact f32 MFMA kernel vs the split-f16 kernel vs the two-launch planes form
(conversion pass + LDS-DMA fed GEMM; GCPX_GEMM_PLANES_CFG=1..4 forces a tile configuration): python tools/time_gemm_split.py [rows ...]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_gcp_amd import runtime as rt, packing as pk
lib = rt.load_library(os.environ.get("GCPX_LIB"))
dev = torch.device("cuda")
def run(name, M, N, K, nb=1, lstm=False):
    x = torch.randn(nb, M, K, device=dev)
    w = torch.randn(nb, N, K) / K ** 0.5
    wp = torch.stack([pk.pack_gemm(w[i]) for i in range(nb)]).contiguous().to(dev)
    packs = [pk.pack_gemm_split(w[i]) for i in range(nb)]
    ws = torch.stack([p[0] for p in packs]).contiguous().to(dev)
    eb = torch.tensor([p[1] for p in packs], dtype=torch.int32, device=dev)
    b = torch.zeros(nb, N, device=dev)
    out = torch.empty(nb, M, N, device=dev)
    H = N // 4
    c, ho, co = (torch.zeros(M, H, device=dev) for _ in range(3))
    for split in (False, True, "planes"):
        a = rt.GemmArgs()
        s = a.src[0]; s.ptr, s.sb, s.sr, s.width = x.data_ptr(), 0, K, K
        a.nsrc, a.M, a.N, a.K, a.rpb = 1, M, N, K, M
        a.wpk, a.bias, a.out, a.ob, a.orow = wp.data_ptr(), b.data_ptr(), out.data_ptr(), 0, N
        if nb > 1:
            a.nbatch, a.z_src_off, a.z_w_off, a.z_bias_off, a.z_out_off = nb, M * K, N * K, N, M * N
        if lstm:
            a.epi = rt.EPI_LSTM
            a.c_prev, a.c_prev_stride, a.h_out, a.c_out, a.hb, a.hrow = c.data_ptr(), H, ho.data_ptr(), co.data_ptr(), 0, H
        if split:
            a.wpk_split, a.w_split_log2_dev = ws.data_ptr(), eb.data_ptr()
        if split == "planes":
            nbytes, nexp = C.c_int64(), C.c_int64()
            a.nbatch = nb if nb > 1 else 0
            rt.check(lib.gcpx_gemm_planes_workspace(M, K, nb, C.byref(nbytes), C.byref(nexp)), "ws")
            planes = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
            exps = torch.empty(nexp.value, dtype=torch.int32, device=dev)
            a.x_planes, a.x_exp, a.x_planes_bytes = planes.data_ptr(), exps.data_ptr(), nbytes.value
        stq = torch.cuda.Stream()
        with torch.cuda.stream(stq):
            for _ in range(3):
                rt.check(lib.gcpx_gemm(C.byref(a), stq.cuda_stream), name)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stq)
            for _ in range(10):
                rt.check(lib.gcpx_gemm(C.byref(a), stq.cuda_stream), name)
            e1.record(stq)
            stq.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"{name:22s} {('planes' if split == 'planes' else 'split ') if split else 'f32   '} M={M:6d} N={N:5d} K={K:5d} nb={nb}: {ms*1e3:8.1f} us  {2.0*M*N*K*nb/ms/1e9:7.1f} TF")
Ms = [int(v) for v in sys.argv[1:]] or [128, 256, 512, 1024, 32768]
for M in Ms:
    run("lstm", M, 2048, 1024, lstm=True)
    if len(Ms) > 1:
        run("merge", M, 512, 1024, nb=6)
if len(Ms) > 1:
    run("lstm0 fused", 1024, 2048, 1280, lstm=True)
