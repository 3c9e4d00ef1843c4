"""Run-to-run check of the training head's gradient rows (GCPX_HEAD_DLM_NLL_GRAD) at the c2 training shapes: `reps` launches of one
problem, every launch compared bit for bit with the first; prints how many values differ, in which lanes of the 16-byte stores (the
slot within the 112-slot row), and what the differing values look like.  GCPX_LIB=<variant build> python tools/head_grad_determinism.py [reps]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_gcp_amd import runtime as rt, packing as pk
lib = rt.load_library(os.environ.get("GCPX_LIB") or None)
dev = torch.device("cuda")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
Fr, S, R = 2304, 64, 1408
torch.manual_seed(0)
x = torch.randn(Fr, S, S, 16, device=dev)
sc, sh = torch.rand(16, device=dev) + 0.5, torch.randn(16, device=dev) * 0.2
w, b = torch.randn(100, 16, 3, 3) / 20.0, torch.randn(100) * 0.1
perm = pk.dlm_channel_perm(10)
wp = pk.pack_dlm_head(w, perm).to(dev)
ws, e = pk.pack_conv3x3_split(w, perm)
ws = ws.to(dev)
permt = torch.tensor(perm)
bk = torch.zeros(len(perm)); bk[permt >= 0] = b[permt[permt >= 0]]; bk = bk.to(dev)
img = torch.zeros(Fr, 3, S, S, device=dev)
rows = torch.full((Fr,), -1, dtype=torch.int32)
rows[torch.randperm(Fr)[:R]] = torch.arange(R, dtype=torch.int32)
rows = rows.to(dev)
grad = torch.empty(R, S, S, len(perm), device=dev)
tgt = torch.rand(R, 3, S, S, device=dev) * 2 - 1
part = torch.zeros(64, R, device=dev)
wgt = torch.ones(R, device=dev)
a = rt.ConvArgs()
s = a.src[0]; s.ptr, s.C, s.frame_div, s.act, s.scale, s.shift = x.data_ptr(), 16, 1, rt.ACT_LRELU, sc.data_ptr(), sh.data_ptr()
a.nsrc, a.F, a.Hin, a.Win, a.Hout, a.Wout, a.Cin, a.Cout, a.out_pitch = 1, Fr, S, S, S, S, 16, 100, len(perm)
a.wpk, a.bias, a.images, a.out, a.head_mode = wp.data_ptr(), bk.data_ptr(), img.data_ptr(), grad.data_ptr(), rt.HEAD_DLM_NLL_GRAD
a.wpk_split, a.w_split_log2, a.raw_row_map = ws.data_ptr(), e, rows.data_ptr()
a.nll_target, a.nll_partial, a.nll_rows, a.nll_row_weight, a.nll_scale = tgt.data_ptr(), part.data_ptr(), R, wgt.data_ptr(), 1e-3
st = torch.cuda.current_stream().cuda_stream
first, total = None, 0
for rep in range(reps):
    grad.fill_(float("nan"))
    rt.check(lib.gcpx_conv3x3(C.byref(a), st), "head")
    torch.cuda.synchronize()
    if first is None:
        first = grad.clone()
        continue
    d = (grad != first) & ~(torch.isnan(grad) & torch.isnan(first))
    n = int(d.sum())
    total += n
    if n:
        idx = torch.nonzero(d)
        slots = torch.bincount(idx[:, 3], minlength=len(perm)).tolist()
        cols = torch.bincount(idx[:, 2] % 16, minlength=16).tolist()
        vals = grad[d][:6].tolist(); ref = first[d][:6].tolist()
        print(f"launch {rep + 1}: {n} of {grad.numel()} values differ; by slot {[(i, c) for i, c in enumerate(slots) if c]}; by pixel column mod 16 {cols}; got {vals} first launch {ref}")
print(f"{os.environ.get('GCPX_LIB') or 'libgcpx.so'}: {total} differing values over {reps - 1} repeats of {grad.numel()} values")
