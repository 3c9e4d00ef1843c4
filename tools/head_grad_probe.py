"""Where the training head's gradient rows (GCPX_HEAD_DLM_NLL_GRAD) differ most from float64 autograd of the oracle's likelihood: per
parameter kind (logit / mean / log-scale / coefficient) the largest |error| relative to the row's largest gradient, and the worst
element's context.  GCPX_LIB=<other build> compares builds."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from video_gcp_amd import runtime as rt, packing as pk, config
from oracle import gcp_model_oracle as O
lib = rt.load_library(os.environ.get("GCPX_LIB") or None)
dev = torch.device("cuda")
hp = config("c1")
torch.manual_seed(23)
S, Fr, R = 64, 24, 16
x = torch.randn(Fr, S, S, 16)
sc, sh = torch.rand(16) + 0.5, torch.randn(16) * 0.2
w, b = torch.randn(100, 16, 3, 3) / 20.0, torch.randn(100) * 0.1
tgt = torch.rand(R, 3, S, S) * 2 - 1
rows = torch.full((Fr,), -1, dtype=torch.int32)
sel = torch.randperm(Fr)[:R]
rows[sel] = torch.arange(R, dtype=torch.int32)
wgt = torch.ones(R)
scale = 1e-3
perm = pk.dlm_channel_perm(10)
permt = torch.tensor(perm)
wp = pk.pack_dlm_head(w, perm).to(dev)
ws, e = pk.pack_conv3x3_split(w, perm)
ws = ws.to(dev)
bk = torch.zeros(len(perm)); bk[permt >= 0] = b[permt[permt >= 0]]
xd, rd, td, wd = x.to(dev), rows.to(dev), tgt.to(dev), wgt.to(dev)
nit = (S // 4) * (S // 16)
part = torch.zeros(nit, R, device=dev)
img = torch.zeros(Fr, 3, S, S, device=dev)
grad = torch.zeros(R, S, S, len(perm), device=dev)
scd, shd, bkd = sc.to(dev), sh.to(dev), bk.to(dev)
a = rt.ConvArgs()
s = a.src[0]; s.ptr, s.C, s.frame_div, s.act, s.scale, s.shift = xd.data_ptr(), 16, 1, rt.ACT_LRELU, scd.data_ptr(), shd.data_ptr()
a.nsrc, a.F, a.Hin, a.Win, a.Hout, a.Wout, a.Cin, a.Cout, a.out_pitch = 1, Fr, S, S, S, S, 16, 100, len(perm)
a.wpk, a.bias, a.images, a.out, a.head_mode = wp.data_ptr(), bkd.data_ptr(), img.data_ptr(), grad.data_ptr(), rt.HEAD_DLM_NLL_GRAD
a.wpk_split, a.w_split_log2, a.raw_row_map = ws.data_ptr(), e, rd.data_ptr()
a.nll_target, a.nll_partial, a.nll_rows, a.nll_row_weight, a.nll_scale = td.data_ptr(), part.data_ptr(), R, wd.data_ptr(), scale
rt.check(lib.gcpx_conv3x3(C.byref(a), torch.cuda.current_stream().cuda_stream), "head")
torch.cuda.synchronize()
slots = torch.nonzero(permt >= 0)[:, 0]
inv = torch.empty(100, dtype=torch.long); inv[permt[slots]] = slots
frame_of = {int(r): f for f, r in enumerate(rows.tolist()) if r >= 0}
kinds = {"logit": range(0, 10), "mean": range(10, 40), "log_scale": range(40, 70), "coeff": range(70, 100)}   # PixelCNN++ order within l[:, nm:]: [3][3 nm]
# channel c of the head: c < 10 logits; then for colour k: means 10+30k .. +9, log-scales 20+30k .. +9, coefficients 30+30k .. +9
def kind(c):
    if c < 10: return "logit"
    return ("mean", "log_scale", "coeff")[((c - 10) % 30) // 10]
worst = {}
for r in range(R):
    f = frame_of[r]
    xin = F.leaky_relu(x[f].permute(2, 0, 1)[None].double() * sc[None, :, None, None].double() + sh[None, :, None, None].double(), 0.2)
    head = F.conv2d(xin, w.double(), b.double(), padding=1).requires_grad_(True)
    nll = O.dlm_nll(head, tgt[[r]].double(), hp).sum()
    (g,) = torch.autograd.grad(nll * scale, head)
    got = grad[r].cpu().index_select(-1, inv).permute(2, 0, 1).double()
    err = (got - g[0]).abs() / float(g.abs().max())
    for c in range(100):
        k = kind(c)
        m = float(err[c].max())
        if m > worst.get(k, (0,))[0]:
            i = int(err[c].argmax()); y, xx = i // S, i % S
            worst[k] = (m, r, c, y, xx, float(got[c, y, xx]), float(g[0, c, y, xx]), float(head[0, c, y, xx]))
for k, v in worst.items():
    print(f"{k:10s} worst |err| / row max = {v[0]:.3e}  row {v[1]} channel {v[2]} pixel ({v[3]},{v[4]}) got {v[5]:.6e} want {v[6]:.6e} head value {v[7]:.4f}")
