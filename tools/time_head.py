"""Time the output-head conv in its different epilogue modes (experiment: how much of the launch is epilogue VALU)."""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_gcp_amd import runtime as rt, packing as pk
lib = rt.load_library()
dev = torch.device("cuda")
Fr, S = 2032, 64
x = torch.randn(Fr, S, S, 16, device=dev)
sc, sh = torch.rand(16, device=dev) + 0.5, torch.randn(16, device=dev) * 0.2
w, b = torch.randn(100, 16, 3, 3) / 12.0, torch.randn(100) * 0.1
perm = pk.dlm_channel_perm(10)
wp = pk.pack_dlm_head(w, perm).to(dev)
permt = torch.tensor(perm)
bk = torch.zeros(len(perm)); bk[permt >= 0] = b[permt[permt >= 0]]; bk = bk.to(dev)
img = torch.zeros(Fr, 3, S, S, device=dev)
a = rt.ConvArgs()
s = a.src[0]; s.ptr, s.C, s.frame_div, s.act, s.scale, s.shift = x.data_ptr(), 16, 1, rt.ACT_LRELU, sc.data_ptr(), sh.data_ptr()
a.nsrc, a.F, a.Hin, a.Win, a.Hout, a.Wout, a.Cin, a.Cout, a.out_pitch = 1, Fr, S, S, S, S, 16, 100, len(perm)
a.wpk, a.bias, a.images = wp.data_ptr(), bk.data_ptr(), img.data_ptr()
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for mode, name in [(rt.HEAD_DLM_MEAN, "dlm_mean"), (rt.HEAD_TANH_NCHW, "tanh3 (tiny epilogue)"), (rt.HEAD_DLM_MEAN, "dlm_mean")]:
        a.head_mode = mode
        for _ in range(3):
            rt.check(lib.gcpx_conv3x3(C.byref(a), st.cuda_stream), "head")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(10):
            rt.check(lib.gcpx_conv3x3(C.byref(a), st.cuda_stream), "head")
        e1.record(st)
        st.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"{name}: {ms:.3f} ms  -> {2*64*64*100*16*9*Fr/ms/1e9:.1f} TF algorithmic")
