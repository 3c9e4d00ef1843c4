"""Time the decoder upsample+conv block `additional_conv_layer` (addl) standalone at c2 size; ablation aid."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_gcp_amd import runtime as rt, packing as pk
lib = rt.load_library(os.environ.get("GCPX_LIB"))
dev = torch.device("cuda")
def run(name, Fr, Hin, c_prev, c_skip, cout, nodes, split=False):
    x = torch.randn(Fr, Hin, Hin, c_prev, device=dev)
    sc, sh = torch.rand(c_prev, device=dev) + 0.5, torch.randn(c_prev, device=dev) * 0.2
    a = rt.ConvArgs()
    s = a.src[0]; s.ptr, s.C, s.frame_div, s.act, s.scale, s.shift = x.data_ptr(), c_prev, 1, rt.ACT_LRELU, sc.data_ptr(), sh.data_ptr()
    n = 1
    if c_skip:
        sk = torch.randn(Fr // nodes, Hin, Hin, c_skip, device=dev)
        s1 = a.src[1]; s1.ptr, s1.C, s1.frame_div, s1.act = sk.data_ptr(), c_skip, nodes, rt.ACT_NONE
        n = 2
    cin = c_prev + c_skip
    w = torch.randn(cout, cin, 3, 3) / (9 * cin) ** 0.5
    wp, b = pk.pack_conv3x3(w, 16 if cout == 16 else 32).to(dev), torch.zeros((cout + 15) // 16 * 16, device=dev)
    out = torch.empty(Fr, 2 * Hin, 2 * Hin, cout, device=dev)
    a.nsrc, a.F, a.Hin, a.Win, a.Hout, a.Wout, a.Cin, a.Cout, a.out_pitch, a.upsample = n, Fr, Hin, Hin, 2 * Hin, 2 * Hin, cin, cout, cout, 1
    a.wpk, a.bias, a.out, a.stats_partial = wp.data_ptr(), b.data_ptr(), out.data_ptr(), out.data_ptr()
    if split == "fold":
        if not (cout == 16 and cin == 32):
            return
        ws, e = pk.pack_conv3x3_fold(w)
        ws = ws.to(dev)
        a.wpk_split, a.w_split_log2, a.split_layout = ws.data_ptr(), e, rt.SPLIT_ROWFOLD
    elif split:
        ws, e = pk.pack_conv3x3_split(w) if cout == 16 else pk.pack_conv3x3_split32(w)
        ws = ws.to(dev)
        a.wpk_split, a.w_split_log2 = ws.data_ptr(), e
    G = lib.gcpx_conv3x3_grid(C.byref(a))
    st = torch.zeros(G, 2, (cout + 15) // 16 * 16, device=dev)
    a.stats_partial = st.data_ptr()
    stq = torch.cuda.Stream()
    with torch.cuda.stream(stq):
        for _ in range(3):
            rt.check(lib.gcpx_conv3x3(C.byref(a), stq.cuda_stream), name)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stq)
        for _ in range(10):
            rt.check(lib.gcpx_conv3x3(C.byref(a), stq.cuda_stream), name)
        e1.record(stq)
        stq.synchronize()
    ms = e0.elapsed_time(e1) / 10
    fl = 2.0 * (2 * Hin) ** 2 * cout * cin * 9 * Fr
    print(f"{name:12s} {ms*1e3:8.1f} us  {fl/ms/1e9:6.1f} TF")
only = sys.argv[1:]            # e.g. "addl fold" "pyr0 fold": run just these
_run = run
def run(name, *args):
    if not only or name in only:
        _run(name, *args)
for split in (False, True, "fold"):
    tag = " fold" if split == "fold" else " split" if split else ""
    run("addl" + tag, 2032, 32, 16, 16, 16, 127, split)
    run("pyr0" + tag, 2032, 16, 32, 0, 16, 1, split)
    run("pyr1" + tag, 2032, 8, 64, 64, 32, 127, split)
    run("pyr2" + tag, 2032, 4, 128, 0, 64, 1, split)
