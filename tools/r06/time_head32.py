"""Standalone timing of the 32x32-tile head (csrc/conv3x3_head32.hip) at the c2 shapes (2032 node frames, 1280 matched) next to the round-5
kernel: mean only, fused likelihood, likelihood + gradient.  GCPX_LIB selects a build (tools/build_variant.sh)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
ws, e = pk.pack_conv3x3_split(w, perm); ws = ws.to(dev)
ws32, e32 = pk.pack_head32_split(w, perm); ws32 = ws32.to(dev)
permt = torch.tensor(perm)
bk = torch.zeros(len(perm)); bk[permt >= 0] = b[permt[permt >= 0]]; bk = bk.to(dev)
img = torch.zeros(Fr, 3, S, S, device=dev)
rows = torch.full((Fr,), -1, dtype=torch.int32)
# the forward's matching: frames of a sequence alternate matched / unmatched runs; a random selection is the harder mix
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
a.nll_target, a.nll_partial, a.nll_rows, a.nll_row_weight, a.nll_scale = tgt.data_ptr(), part.data_ptr(), R, wgt.data_ptr(), 1e-3
phase = torch.zeros(4096, 4, device=dev)
if os.environ.get("HEAD32_TIMING"):
    a.stats_partial = phase.data_ptr()
st = torch.cuda.Stream()
only = os.environ.get("HEAD_MODES", "mean,nll,grad").split(",")
with torch.cuda.stream(st):
    for layout, lname in [(rt.SPLIT_HEAD32, "32x32 head"), (rt.SPLIT_PLAIN, "round-5 head")]:
        if layout == rt.SPLIT_PLAIN and os.environ.get("HEAD32_ONLY"):
            continue
        a.wpk_split, a.w_split_log2, a.split_layout = (ws32.data_ptr(), e32, layout) if layout == rt.SPLIT_HEAD32 else (ws.data_ptr(), e, layout)
        for mode, name, out, rm in [(rt.HEAD_DLM_MEAN, "mean", None, None), (rt.HEAD_DLM_NLL, "nll", None, rows), (rt.HEAD_DLM_NLL_GRAD, "grad", raw, rows)]:
            if name not in only:
                continue
            a.head_mode, a.out, a.raw_row_map = mode, (out.data_ptr() if out is not None else None), (rm.data_ptr() if rm is not None else None)
            ts = []
            for rep in range(3):
                for _ in range(2):
                    rt.check(lib.gcpx_conv3x3(C.byref(a), st.cuda_stream), "head")
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                for _ in range(5):
                    rt.check(lib.gcpx_conv3x3(C.byref(a), st.cuda_stream), "head")
                e1.record(st)
                st.synchronize()
                ts.append(e0.elapsed_time(e1) / 5)
            print(f"{os.environ.get('GCPX_LIB', 'this build')[-28:]:28s} {lname:14s} {name:5s} {min(ts):.3f} ms")
            if os.environ.get("HEAD32_TIMING") and layout == rt.SPLIT_HEAD32:
                ph = phase.cpu()
                ph = ph[ph[:, 3] > 0]
                it = ph[:, 3].sum()
                print(f"      s_memtime ticks per item (mean over {len(ph)} wavefronts): MFMA passes {ph[:, 0].sum() / it:.0f}, epilogues {ph[:, 1].sum() / it:.0f}, "
                      f"rest {ph[:, 2].sum() / it:.0f}; per wavefront total {float((ph[:, 0] + ph[:, 1] + ph[:, 2]).mean()):.0f} ticks for {float(ph[:, 3].mean()):.1f} items")
