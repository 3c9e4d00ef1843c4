"""Time the encoder 4x4 stride-2 blocks at the c2 size (1280 frames), exact f32 vs split-f16: python tools/time_enc.py"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_gcp_amd import runtime as rt, packing as pk
lib = rt.load_library(os.environ.get("GCPX_LIB"))
dev = torch.device("cuda")
def run(name, Fr, Hin, cin, cout):
    x = torch.randn(Fr, Hin, Hin, cin, device=dev)
    sc, sh = torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.2
    w = torch.randn(cout, cin, 4, 4) / (16 * cin) ** 0.5
    wp, b = pk.pack_conv4x4(w).to(dev), torch.zeros(cout, device=dev)
    ws, e = pk.pack_conv4x4_split(w)
    ws = ws.to(dev)
    out = torch.empty(Fr, Hin // 2, Hin // 2, cout, device=dev)
    st = torch.zeros(lib.gcpx_conv4x4s2_grid(), 2, cout, device=dev)
    for split in (False, True):
        a = rt.ConvArgs()
        s = a.src[0]; s.ptr, s.C, s.frame_div, s.act, s.scale, s.shift = x.data_ptr(), cin, 1, rt.ACT_LRELU, sc.data_ptr(), sh.data_ptr()
        a.nsrc, a.F, a.Hin, a.Win, a.Hout, a.Wout, a.Cin, a.Cout, a.out_pitch = 1, Fr, Hin, Hin, Hin // 2, Hin // 2, cin, cout, cout
        a.wpk, a.bias, a.out, a.stats_partial = wp.data_ptr(), b.data_ptr(), out.data_ptr(), st.data_ptr()
        if split:
            a.wpk_split, a.w_split_log2 = ws.data_ptr(), e
        stq = torch.cuda.Stream()
        with torch.cuda.stream(stq):
            for _ in range(2):
                rt.check(lib.gcpx_conv4x4s2(C.byref(a), stq.cuda_stream), name)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stq)
            for _ in range(5):
                rt.check(lib.gcpx_conv4x4s2(C.byref(a), stq.cuda_stream), name)
            e1.record(stq)
            stq.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"{name:10s} {'split' if split else 'f32  '} {ms*1e3:8.1f} us  {2.0*(Hin//2)**2*cout*cin*16*Fr/ms/1e9:6.1f} TF")
run("enc1", 1280, 32, 16, 32)
run("enc2", 1280, 16, 32, 64)
run("enc3", 1280, 8, 64, 128)
run("enc1 x16", 16, 32, 16, 32)
