"""Standalone timing of the split-f16 output head in its modes at the c2 shapes (2032 node frames, 1280 of them matched): mean only,
stored parameters (BOTH), fused likelihood (NLL), fused likelihood + gradient (NLL_GRAD).  A/B aid: GCPX_HEAD_NO_PINGPONG=1."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_gcp_amd import runtime as rt, packing as pk
lib = rt.load_library()
dev = torch.device("cuda")
Fr, S, R = 2032, 64, 1280
torch.manual_seed(0)
x = torch.randn(Fr, S, S, 16, device=dev)
sc, sh = torch.rand(16, device=dev) + 0.5, torch.randn(16, device=dev) * 0.2
w, b = torch.randn(100, 16, 3, 3) / 12.0, torch.randn(100) * 0.1
perm = pk.dlm_channel_perm(10)
wp = pk.pack_dlm_head(w, perm).to(dev)
ws, e = pk.pack_conv3x3_split(w, perm)
ws = ws.to(dev)
permt = torch.tensor(perm)
bk = torch.zeros(len(perm)); bk[permt >= 0] = b[permt[permt >= 0]]; bk = bk.to(dev)
img = torch.zeros(Fr, 3, S, S, device=dev)
rows = torch.full((Fr,), -1, dtype=torch.int32)
sel = torch.randperm(Fr)[:R]
rows[sel] = torch.arange(R, dtype=torch.int32)
rows = rows.to(dev)
raw = torch.empty(R, S, S, len(perm), device=dev)
tgt = torch.rand(R, 3, S, S, device=dev) * 2 - 1
part = torch.zeros(64, R, device=dev)
wgt = torch.ones(R, device=dev)
a = rt.ConvArgs()
s = a.src[0]; s.ptr, s.C, s.frame_div, s.act, s.scale, s.shift = x.data_ptr(), 16, 1, rt.ACT_LRELU, sc.data_ptr(), sh.data_ptr()
a.nsrc, a.F, a.Hin, a.Win, a.Hout, a.Wout, a.Cin, a.Cout, a.out_pitch = 1, Fr, S, S, S, S, 16, 100, len(perm)
a.wpk, a.bias, a.images = wp.data_ptr(), bk.data_ptr(), img.data_ptr()
a.wpk_split, a.w_split_log2 = ws.data_ptr(), e
a.nll_target, a.nll_partial, a.nll_rows, a.nll_row_weight, a.nll_scale = tgt.data_ptr(), part.data_ptr(), R, wgt.data_ptr(), 1e-3
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for mode, name, out, rm in [(rt.HEAD_DLM_MEAN, "mean only", None, None), (rt.HEAD_DLM_BOTH, "stored parameters", raw, rows),
                                (rt.HEAD_DLM_NLL, "fused likelihood", None, rows), (rt.HEAD_DLM_NLL_GRAD, "likelihood + gradient", raw, rows)]:
        a.head_mode, a.out, a.raw_row_map = mode, (out.data_ptr() if out is not None else None), (rm.data_ptr() if rm is not None else None)
        ts = []
        for rep in range(4):
            for _ in range(2):
                rt.check(lib.gcpx_conv3x3(C.byref(a), st.cuda_stream), "head")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(5):
                rt.check(lib.gcpx_conv3x3(C.byref(a), st.cuda_stream), "head")
            e1.record(st)
            st.synchronize()
            ts.append(e0.elapsed_time(e1) / 5)
        print(f"{name:24s} {min(ts):.3f} ms (min of 4 x 5 launches; back-to-back launches run hotter than inside the forward)")
