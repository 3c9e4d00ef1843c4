"""Split-f16 output head vs the exact-f32 head: error of both against a float64 conv, and launch times at the c2 size.
Run on the GPU box: python tools/head_split_check.py [F_timing]"""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import video_gcp_amd as V
from video_gcp_amd import runtime as rt, packing as pk

lib = rt.load_library()
dev = "cuda"


def args(xd, sc, sh, Fr, S, wp, bk, raw, img, mode, split=None, e=0):
    a = rt.ConvArgs()
    s = a.src[0]
    s.ptr, s.C, s.frame_div, s.act = xd.data_ptr(), 16, 1, rt.ACT_LRELU
    s.scale, s.shift = sc.data_ptr(), sh.data_ptr()
    a.nsrc, a.Cin, a.F, a.Hin, a.Win, a.Hout, a.Wout, a.Cout, a.out_pitch = 1, 16, Fr, S, S, S, S, 100, 100
    a.upsample, a.head_mode, a.wpk, a.bias = 0, mode, wp.data_ptr(), bk.data_ptr()
    a.out = raw.data_ptr() if raw is not None else None
    a.images = img.data_ptr()
    if split is not None:
        a.wpk_split, a.w_split_log2 = split.data_ptr(), e
    return a


def main():
    torch.manual_seed(0)
    S, Fr = 64, 4
    perm = pk.dlm_channel_perm(10)
    permt = torch.tensor(perm)
    for amp in ((1.0,) if os.environ.get('TIMING_ONLY') else (1.0, 1e-3, 300.0)):
        x = torch.randn(Fr, 16, S, S) * amp
        sc, sh = torch.rand(16) + 0.5, torch.randn(16) * 0.2 * amp
        w, b = torch.randn(100, 16, 3, 3) / 12.0, torch.randn(100) * 0.1
        xin = F.leaky_relu(x * sc[None, :, None, None] + sh[None, :, None, None], 0.2)           # f32, as the kernels compute it
        ref = F.conv2d(xin.double(), w.double(), b.double(), padding=1)
        wp = pk.pack_dlm_head(w, perm).to(dev)
        ws, e = pk.pack_conv3x3_split(w, perm)
        ws = ws.to(dev)
        bk = torch.zeros(len(perm)); bk[permt >= 0] = b[permt[permt >= 0]]; bk = bk.to(dev)
        xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
        scd, shd = sc.to(dev), sh.to(dev)
        inv = torch.empty(100, dtype=torch.long); sl = torch.nonzero(permt >= 0)[:, 0]; inv[permt[sl]] = sl
        res = {}
        for name, split in (("f32", None), ("split", ws)):
            raw = torch.full((Fr, S, S, 100), float("nan"), device=dev)
            img = torch.full((Fr, 3, S, S), float("nan"), device=dev)
            a = args(xd, scd, shd, Fr, S, wp, bk, raw, img, rt.HEAD_DLM_BOTH, split, e)
            rt.check(lib.gcpx_conv3x3(C.byref(a), torch.cuda.current_stream().cuda_stream), name)
            torch.cuda.synchronize()
            got = raw.cpu().index_select(-1, inv).permute(0, 3, 1, 2).double()
            err = (got - ref).abs()
            res[name] = (got, img.cpu())
            print(f"amp {amp:g} {name:6s} raw: max abs err {err.max():.3e}  rms err {err.pow(2).mean().sqrt():.3e}  (rms of output {ref.pow(2).mean().sqrt():.3e})"
                  f"  finite images {bool(torch.isfinite(img).all())}")
        d = (res["f32"][1] - res["split"][1]).abs().max()
        print(f"amp {amp:g} images f32 vs split: max abs diff {d:.3e}")
    # timing at the c2 size
    Fr = int(sys.argv[1]) if len(sys.argv) > 1 else 2032
    xd = torch.randn(Fr, S, S, 16, device=dev)
    img = torch.empty(Fr, 3, S, S, device=dev)
    for name, split in ((("split", ws),) if os.environ.get('TIMING_ONLY') else (("f32", None), ("split", ws))):
        a = args(xd, scd, shd, Fr, S, wp, bk, None, img, rt.HEAD_DLM_MEAN, split, e)
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(3):
            rt.check(lib.gcpx_conv3x3(C.byref(a), st), name)
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(10):
            rt.check(lib.gcpx_conv3x3(C.byref(a), st), name)
        t1.record(); torch.cuda.synchronize()
        print(f"{name:6s} head, F={Fr}: {t0.elapsed_time(t1) / 10 * 1e3:.1f} us / launch")


main()
