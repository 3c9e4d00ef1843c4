"""-m gpu: each C-ABI entry point against a plain PyTorch fp32 (CPU) reference of the same op.
Tolerances are fp32-roundoff class (different summation order only)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from helpers import assert_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    from video_gcp_amd import runtime as rt, packing as pk
    lib = rt.load_library()
    assert torch.cuda.is_available()
    return rt, pk, lib, torch.device("cuda")


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _rowsrc(rt, t, sb, sr, width, **kw):
    s = rt.RowSrc()
    s.ptr, s.sb, s.sr, s.width = t.data_ptr() if hasattr(t, "data_ptr") else t, sb, sr, width
    for k, v in kw.items():
        setattr(s, k, v.data_ptr() if hasattr(v, "data_ptr") else v)
    return s


@pytest.mark.parametrize("M,N,K", [(16, 128, 256), (70, 512, 128), (200, 2048, 1024), (1024, 64, 64)])
def test_gemm_plain(env, M, N, K):
    rt, pk, lib, dev = env
    torch.manual_seed(M + N)
    x, w, b = torch.randn(M, K), torch.randn(N, K) / K ** 0.5, torch.randn(N)
    want = F.linear(x, w, b)
    xd, wp, bd = x.to(dev), pk.pack_gemm(w).to(dev), b.to(dev)
    out = torch.full((M, N), float("nan"), device=dev)
    a = rt.GemmArgs()
    a.src[0] = _rowsrc(rt, xd, 0, K, K)
    a.nsrc, a.M, a.N, a.K, a.rpb = 1, M, N, K, M
    a.wpk, a.bias, a.out, a.ob, a.orow = wp.data_ptr(), bd.data_ptr(), out.data_ptr(), 0, N
    rt.check(lib.gcpx_gemm(C.byref(a), _stream()), "gemm")
    torch.cuda.synchronize()
    assert_close(out, want, atol=2e-5, rtol=1e-5, name="gemm")


@pytest.mark.parametrize("M,rpb,H,N,K", [(16, 1, 128, 256, 512), (48, 4, 64, 64, 256), (300, 4, 256, 256, 1024), (1024, 64, 128, 128, 512)])
def test_gemm_lstm_bwd_epilogue_equals_separate_launch(env, M, rpb, H, N, K):
    """gcpx_gemm_args.lstm_bwd: the cell backward of the layer a data-gradient GEMM feeds, run in the GEMM's epilogue, writes what
    gcpx_gemm followed by gcpx_lstm_bwd (reading the GEMM's output as dh_dense) writes — dgates and dc_prev bit for bit (the same
    device function on the same values) — over the one-tile split-K, multi-tile and one-wavefront-per-block launch forms, a [dx | dh]
    GEMM twice as wide as the cell (columns >= H are plain outputs), rows addressed as (b, j), a dc buffer that is both dc_pos and dc_prev."""
    rt, pk, lib, dev = env
    torch.manual_seed(M + N + K)
    x, w = torch.randn(M, K, device=dev), (torch.randn(N, K) / K ** 0.5)
    wp = pk.pack_gemm(w).to(dev)
    nb = M // rpb
    PB, PR_ = rpb * 2 * H + 32, 2 * H            # stored states: row (b, j) at b*PB + j*PR_ (pitch 2H: [h | c] as in the VRNN)
    gates = torch.rand(M, H, 4, device=dev) * 0.9 + 0.05
    gates[:, :, 2] = gates[:, :, 2] * 2 - 1
    c_prev, c_new = torch.randn(M, 3 * H, device=dev), torch.randn(nb * PB, device=dev)
    dh_pos = torch.randn(nb * PB, device=dev)
    outs = []
    for fused in (False, True):
        out = torch.full((M, N), float("nan"), device=dev)
        dgates = torch.full((M, 4 * H), float("nan"), device=dev)
        dc = torch.arange(nb * PB, device=dev, dtype=torch.float32).sin()           # dc_pos, overwritten in place by dc_prev
        L = rt.LstmBwdArgs()
        L.gates, L.c_prev, L.c_prev_stride = gates.data_ptr(), c_prev.data_ptr() + 4 * H, 3 * H
        L.c_new, L.pb, L.prow = c_new.data_ptr(), PB, PR_
        L.dh_dense, L.dh_stride = out.data_ptr(), N
        L.dh_pos, L.dc_pos = dh_pos.data_ptr(), dc.data_ptr()
        L.dgates, L.dc_prev, L.dcp_stride = dgates.data_ptr(), dc.data_ptr(), 0
        L.M, L.H, L.rpb = M, H, rpb
        # dc_prev row r at dc_prev + r*dcp_stride must alias dc_pos of the same (b, j): only possible with rpb == 1 or a dense pitch —
        # keep the alias for rpb == 1, a separate buffer otherwise
        if rpb == 1:
            L.dcp_stride = PB
        else:
            dcp = torch.full((M, H), float("nan"), device=dev)
            L.dc_prev, L.dcp_stride = dcp.data_ptr(), H
        Ld = torch.frombuffer(bytearray(bytes(L)), dtype=torch.uint8).to(dev)
        a = rt.GemmArgs()
        a.src[0] = _rowsrc(rt, x, 0, K, K)
        a.nsrc, a.M, a.N, a.K, a.rpb = 1, M, N, K, M
        a.wpk, a.out, a.ob, a.orow = wp.data_ptr(), out.data_ptr(), 0, N
        if fused:
            a.lstm_bwd = Ld.data_ptr()
        rt.check(lib.gcpx_gemm(C.byref(a), _stream()), "gemm")
        if not fused:
            rt.check(lib.gcpx_lstm_bwd(C.byref(L), _stream()), "lstm_bwd")
        torch.cuda.synchronize()
        outs.append((out, dgates, dc if rpb == 1 else dcp))
    for (x0, x1), what in zip(zip(*outs), ("out", "dgates", "dc_prev")):
        assert not torch.isnan(x1).any() or what == "dc_prev", what
        assert torch.equal(x0, x1), what


def test_gemm_sources_shift_affine_stats(env):
    """conv1d-over-time form: three shifted sources, affine+LReLU on load, LReLU epilogue, stats partials."""
    rt, pk, lib, dev = env
    torch.manual_seed(3)
    B, T, Cc, N = 3, 20, 32, 48
    x = torch.randn(B, T, Cc)
    sc, sh = torch.rand(Cc) + 0.5, torch.randn(Cc) * 0.1
    w = torch.randn(N, Cc, 3) / (3 * Cc) ** 0.5
    b = torch.randn(N)
    xa = F.leaky_relu(x * sc + sh, 0.2)
    want = F.conv1d(xa.transpose(1, 2), w, b, padding=1).transpose(1, 2)
    xd, scd, shd = x.to(dev), sc.to(dev), sh.to(dev)
    wp = pk.pack_gemm(w.permute(0, 2, 1).reshape(N, 3 * Cc)).to(dev)
    bd = b.to(dev)
    out = torch.zeros(B * T, N, device=dev)
    a = rt.GemmArgs()
    for i, d in enumerate((-1, 0, 1)):
        a.src[i] = _rowsrc(rt, xd, T * Cc, Cc, Cc, shift=d, scale=scd, shiftv=shd, act=rt.ACT_LRELU, cmod=Cc)
    a.nsrc, a.M, a.N, a.K, a.rpb = 3, B * T, N, 3 * Cc, T
    nrb = lib.gcpx_gemm_row_blocks(B * T, N)
    st = torch.zeros(nrb, 2, N, device=dev)
    a.wpk, a.bias, a.out, a.ob, a.orow = wp.data_ptr(), bd.data_ptr(), out.data_ptr(), T * N, N
    a.stats_partial = st.data_ptr()
    rt.check(lib.gcpx_gemm(C.byref(a), _stream()), "gemm")
    torch.cuda.synchronize()
    assert_close(out.view(B, T, N), want, atol=2e-5, rtol=1e-5, name="conv1d-gemm")
    s = st.sum(0).cpu()
    assert_close(s[0], want.reshape(-1, N).sum(0), atol=1e-3, name="stats sum")
    assert_close(s[1], (want.reshape(-1, N) ** 2).sum(0), atol=1e-3, rtol=1e-5, name="stats sumsq")
    # finalize -> scale/shift of a BatchNorm over those rows
    gamma, beta = torch.rand(N) + 0.5, torch.randn(N)
    gd, btd = gamma.to(dev), beta.to(dev)
    scale, shift = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
    rt.check(lib.gcpx_bn_finalize(st.data_ptr(), nrb, N, N, float(B * T), gd.data_ptr(), btd.data_ptr(),
                                  1e-5, scale.data_ptr(), shift.data_ptr(), None, None, 0.0, None, None, _stream()), "bn_finalize")
    torch.cuda.synchronize()
    flat = want.reshape(-1, N)
    mean, var = flat.mean(0), flat.var(0, unbiased=False)
    wsc = gamma / torch.sqrt(var + 1e-5)
    assert_close(scale, wsc, atol=1e-5, rtol=1e-4, name="bn scale")
    assert_close(shift, beta - mean * wsc, atol=1e-5, rtol=1e-4, name="bn shift")


def test_gemm_lstm_epilogue(env):
    rt, pk, lib, dev = env
    torch.manual_seed(5)
    M, H = 40, 64
    x, h, c = torch.randn(M, H), torch.randn(M, H), torch.randn(M, H)
    cell = torch.nn.LSTMCell(H, H)
    with torch.no_grad():
        h1, c1 = cell(x, (h, c))
    w, b = pk.lstm_gate_interleave(cell.weight_ih.detach(), cell.weight_hh.detach(), cell.bias_ih.detach(), cell.bias_hh.detach())
    wp, bd = pk.pack_gemm(w).to(dev), b.to(dev)
    xd, hd, cd = x.to(dev), h.to(dev), c.to(dev)
    ho, co, hc = (torch.zeros(M, H, device=dev) for _ in range(3))
    a = rt.GemmArgs()
    a.src[0] = _rowsrc(rt, xd, 0, H, H)
    a.src[1] = _rowsrc(rt, hd, 0, H, H)
    a.nsrc, a.M, a.N, a.K, a.rpb = 2, M, 4 * H, 2 * H, M
    a.wpk, a.bias, a.epi = wp.data_ptr(), bd.data_ptr(), rt.EPI_LSTM
    a.c_prev, a.c_prev_stride, a.h_out, a.c_out, a.hb, a.hrow, a.h_copy = cd.data_ptr(), H, ho.data_ptr(), co.data_ptr(), 0, H, hc.data_ptr()
    rt.check(lib.gcpx_gemm(C.byref(a), _stream()), "gemm lstm")
    torch.cuda.synchronize()
    assert_close(ho, h1, atol=1e-5, name="h")
    assert_close(co, c1, atol=1e-5, name="c")
    assert_close(hc, h1, atol=1e-5, name="h_copy")


@pytest.mark.parametrize("case", ["unit", "tiny", "row_scales", "k_growth", "outlier", "zero_rows"])
@pytest.mark.parametrize("M,N,K", [(200, 2048, 1024), (130, 192, 128), (512, 512, 1024), (8200, 1024, 192)])
def test_gemm_split_error_vs_float64(env, case, M, N, K):
    """The split-f16 row GEMM (csrc/gemm_split.hip: two f16 pieces per operand, three f16 MFMAs per product, a per-row power-of-two
    scale that follows the row along K) against float64, next to the exact f32 MFMA kernel: rows of magnitude 1e-6 (gradients) next
    to rows of order 1e3, magnitudes growing 1e4-fold along K (the running scale drops and the sums are rescaled), an outlier, and
    all-zero rows.  Both tile widths (128 / 64 columns), the 128-row / 512-thread form of problems with thousands of rows (8200 x 1024)
    and the masked last row block are covered by the shapes."""
    rt, pk, lib, dev = env
    torch.manual_seed(M + K)
    x = torch.randn(M, K)
    if case == "tiny":
        x *= 1e-6
    elif case == "row_scales":
        x *= torch.logspace(-6, 3, M)[:, None]
    elif case == "k_growth":
        x *= torch.logspace(-2, 2, K)[None, :]
    elif case == "outlier":
        x[3, K // 2] = 3e4
    elif case == "zero_rows":
        x[::3] = 0.0
    w, b = torch.randn(N, K) / K ** 0.5, torch.randn(N)
    ref = F.linear(x.double(), w.double(), b.double())
    xd, wp, bd = x.to(dev), pk.pack_gemm(w).to(dev), b.to(dev)
    ws, e = pk.pack_gemm_split(w)
    ws = ws.to(dev)
    err = {}
    for name in ("f32", "split"):
        out = torch.full((M, N), float("nan"), device=dev)
        a = rt.GemmArgs()
        a.src[0] = _rowsrc(rt, xd, 0, K, K)
        a.nsrc, a.M, a.N, a.K, a.rpb = 1, M, N, K, M
        a.wpk, a.bias, a.out, a.ob, a.orow = wp.data_ptr(), bd.data_ptr(), out.data_ptr(), 0, N
        if name == "split":
            a.wpk_split, a.w_split_log2 = ws.data_ptr(), e
        rt.check(lib.gcpx_gemm(C.byref(a), _stream()), name)
        torch.cuda.synchronize()
        assert torch.isfinite(out).all()
        err[name] = (out.cpu().double() - ref).abs()
    # row by row (rows differ by nine orders of magnitude): rms error within 1.5x the exact kernel's, max within 2x (+ one f32 rounding
    # of the row's largest result)
    rs = ref.abs().amax(1) + 1e-30
    rms = {k: v.pow(2).mean(1).sqrt() for k, v in err.items()}
    assert bool((rms["split"] <= 1.5 * rms["f32"] + 1e-7 * rs).all()), float((rms["split"] / (rms["f32"] + 1e-7 * rs)).max())
    assert bool((err["split"].amax(1) <= 2.0 * err["f32"].amax(1) + 4e-7 * rs).all())


def test_gemm_split_lstm_sources_and_batches(env):
    """split-f16 GEMM with what the tree levels use: several concatenated sources with a row gather and a masked shift, the LSTM-cell
    epilogue (h, c, dense h copy), and the batched form (blockIdx.z problems with their own weights, bias and weight exponent)."""
    rt, pk, lib, dev = env
    torch.manual_seed(9)
    M, H = 300, 128
    pool = torch.randn(500, H)
    ridx = torch.randint(0, 500, (M,), dtype=torch.int32)
    x, h, c = pool[ridx.long()], torch.randn(M, H), torch.randn(M, H)
    cell = torch.nn.LSTMCell(H, H)
    with torch.no_grad():
        h1, c1 = cell(x, (h, c))
    w, b = pk.lstm_gate_interleave(cell.weight_ih.detach(), cell.weight_hh.detach(), cell.bias_ih.detach(), cell.bias_hh.detach())
    wp, bd = pk.pack_gemm(w).to(dev), b.to(dev)
    ws, e = pk.pack_gemm_split(w)
    ws = ws.to(dev)
    pd, rd, hd, cd = pool.to(dev), ridx.to(dev), h.to(dev), c.to(dev)
    ho, co, hc = (torch.full((M, H), float("nan"), device=dev) for _ in range(3))
    a = rt.GemmArgs()
    a.src[0] = _rowsrc(rt, pd, 0, H, H, rowidx=rd)
    a.src[1] = _rowsrc(rt, hd, 0, H, H)
    a.nsrc, a.M, a.N, a.K, a.rpb = 2, M, 4 * H, 2 * H, M
    a.wpk, a.bias, a.epi = wp.data_ptr(), bd.data_ptr(), rt.EPI_LSTM
    a.wpk_split, a.w_split_log2 = ws.data_ptr(), e
    a.c_prev, a.c_prev_stride, a.h_out, a.c_out, a.hb, a.hrow, a.h_copy = cd.data_ptr(), H, ho.data_ptr(), co.data_ptr(), 0, H, hc.data_ptr()
    rt.check(lib.gcpx_gemm(C.byref(a), _stream()), "gemm split lstm")
    torch.cuda.synchronize()
    assert_close(ho, h1, atol=1e-5, name="h")
    assert_close(co, c1, atol=1e-5, name="c")
    assert_close(hc, h1, atol=1e-5, name="h_copy")
    # the same cell over thousands of rows: the 128-row / 512-thread workgroups (gathered source, LSTM epilogue, masked last block)
    Mb = 16500
    ridx_b = torch.randint(0, 500, (Mb,), dtype=torch.int32)
    xb_, hb_, cb_ = pool[ridx_b.long()], torch.randn(Mb, H), torch.randn(Mb, H)
    with torch.no_grad():
        h1b, c1b = cell(xb_, (hb_, cb_))
    rdb, hdb, cdb = ridx_b.to(dev), hb_.to(dev), cb_.to(dev)
    hob, cob = (torch.full((Mb, H), float("nan"), device=dev) for _ in range(2))
    a = rt.GemmArgs()
    a.src[0] = _rowsrc(rt, pd, 0, H, H, rowidx=rdb)
    a.src[1] = _rowsrc(rt, hdb, 0, H, H)
    a.nsrc, a.M, a.N, a.K, a.rpb = 2, Mb, 4 * H, 2 * H, Mb
    a.wpk, a.bias, a.epi = wp.data_ptr(), bd.data_ptr(), rt.EPI_LSTM
    a.wpk_split, a.w_split_log2 = ws.data_ptr(), e
    a.c_prev, a.c_prev_stride, a.h_out, a.c_out, a.hb, a.hrow = cdb.data_ptr(), H, hob.data_ptr(), cob.data_ptr(), 0, H
    rt.check(lib.gcpx_gemm(C.byref(a), _stream()), "gemm split lstm, many rows")
    torch.cuda.synchronize()
    assert_close(hob, h1b, atol=1e-5, name="h, many rows")
    assert_close(cob, c1b, atol=1e-5, name="c, many rows")
    # conv1d-over-time form (three shifted sources with an affine + LReLU on load, rows masked at the sequence ends), LReLU epilogue
    B, T, Cc, N = 4, 40, 64, 128
    xs = torch.randn(B, T, Cc)
    sc, sh = torch.rand(Cc) + 0.5, torch.randn(Cc) * 0.1
    wc, bc = torch.randn(N, Cc, 3) / (3 * Cc) ** 0.5, torch.randn(N)
    want = F.leaky_relu(F.conv1d(F.leaky_relu(xs * sc + sh, 0.2).transpose(1, 2), wc, bc, padding=1).transpose(1, 2), 0.2)
    w2 = wc.permute(0, 2, 1).reshape(N, 3 * Cc)
    xd, scd, shd, bcd = xs.to(dev), sc.to(dev), sh.to(dev), bc.to(dev)
    wp2 = pk.pack_gemm(w2).to(dev)
    ws2, e2 = pk.pack_gemm_split(w2)
    ws2 = ws2.to(dev)
    out = torch.full((B * T, N), float("nan"), device=dev)
    a = rt.GemmArgs()
    for i, d in enumerate((-1, 0, 1)):
        a.src[i] = _rowsrc(rt, xd, T * Cc, Cc, Cc, shift=d, scale=scd, shiftv=shd, act=rt.ACT_LRELU, cmod=Cc)
    a.nsrc, a.M, a.N, a.K, a.rpb = 3, B * T, N, 3 * Cc, T
    a.wpk, a.bias, a.out, a.ob, a.orow, a.epi = wp2.data_ptr(), bcd.data_ptr(), out.data_ptr(), T * N, N, rt.EPI_LRELU
    a.wpk_split, a.w_split_log2 = ws2.data_ptr(), e2
    rt.check(lib.gcpx_gemm(C.byref(a), _stream()), "gemm split conv1d")
    torch.cuda.synchronize()
    assert_close(out.view(B, T, N), want, atol=2e-5, rtol=1e-5, name="conv1d-gemm split")
    # batched: 3 problems, weights of different magnitude (their own exponents)
    nb, M2, N2, K2 = 3, 256, 128, 256
    xb = torch.randn(nb, M2, K2)
    wb = torch.randn(nb, N2, K2) / K2 ** 0.5 * torch.tensor([1.0, 1e-3, 50.0])[:, None, None]
    bb = torch.randn(nb, N2)
    wantb = torch.einsum("bmk,bnk->bmn", xb, wb) + bb[:, None, :]
    packs = [pk.pack_gemm_split(wb[i]) for i in range(nb)]
    wsb = torch.stack([p_[0] for p_ in packs]).contiguous().to(dev)
    eb = torch.tensor([p_[1] for p_ in packs], dtype=torch.int32, device=dev)
    wpb = torch.stack([pk.pack_gemm(wb[i]) for i in range(nb)]).contiguous().to(dev)
    xbd, bbd = xb.to(dev), bb.to(dev)
    outb = torch.full((nb, M2, N2), float("nan"), device=dev)
    a = rt.GemmArgs()
    a.src[0] = _rowsrc(rt, xbd, 0, K2, K2)
    a.nsrc, a.M, a.N, a.K, a.rpb = 1, M2, N2, K2, M2
    a.wpk, a.bias, a.out, a.ob, a.orow = wpb.data_ptr(), bbd.data_ptr(), outb.data_ptr(), 0, N2
    a.nbatch, a.z_src_off, a.z_w_off, a.z_bias_off, a.z_out_off = nb, M2 * K2, N2 * K2, N2, M2 * N2
    a.wpk_split, a.w_split_log2_dev = wsb.data_ptr(), eb.data_ptr()
    rt.check(lib.gcpx_gemm(C.byref(a), _stream()), "gemm split batched")
    torch.cuda.synchronize()
    assert_close(outb, wantb, atol=3e-5, rtol=2e-5, name="batched split")


def _predictor_ref(x, Ws, n_mid, groups=8):
    h = F.leaky_relu(F.linear(x, Ws["w_in"], Ws["b_in"]), 0.2)
    for i in range(n_mid):
        h = F.linear(h, Ws["w_mid"][i], Ws["b_mid"][i])
        h = F.leaky_relu(F.group_norm(h, groups, Ws["g"][i], Ws["be"][i], 1e-5), 0.2)
    return F.linear(h, Ws["w_out"], Ws["b_out"])


@pytest.mark.parametrize("mid,in_dims,out_dim,M,gauss", [(128, (128, 128), 80, 37, False), (128, (128, 128, 128), 512, 50, True),
                                                         (32, (128, 128, 256), 1536, 16, False), (128, (128,), 1, 100, False)])
def test_mlp(env, mid, in_dims, out_dim, M, gauss):
    rt, pk, lib, dev = env
    torch.manual_seed(mid + out_dim)
    n_mid, K = 3, sum(in_dims)
    xs = [torch.randn(M, d) for d in in_dims]
    Ws = dict(w_in=torch.randn(mid, K) / K ** 0.5, b_in=torch.randn(mid) * 0.1,
              w_mid=torch.randn(n_mid, mid, mid) / mid ** 0.5, b_mid=torch.randn(n_mid, mid) * 0.1,
              g=torch.rand(n_mid, mid) + 0.5, be=torch.randn(n_mid, mid) * 0.1,
              w_out=torch.randn(out_dim, mid) / mid ** 0.5, b_out=torch.randn(out_dim) * 0.1)
    want = _predictor_ref(torch.cat(xs, 1), Ws, n_mid)
    out_pad = (out_dim + 15) // 16 * 16
    d = {k: v.to(dev) for k, v in dict(
        w_in=pk.pack_gemm(Ws["w_in"]), b_in=Ws["b_in"], w_mid=torch.stack([pk.pack_gemm(Ws["w_mid"][i]) for i in range(n_mid)]),
        b_mid=Ws["b_mid"], g=Ws["g"], be=Ws["be"], w_out=pk.pack_gemm(Ws["w_out"]), b_out=pk.pad_vec(Ws["b_out"], out_pad)).items()}
    xd = [x.to(dev) for x in xs]
    out = torch.full((M, out_dim), float("nan"), device=dev)
    a = rt.MlpArgs()
    for i, x in enumerate(xd):
        a.src[i] = _rowsrc(rt, x, 0, in_dims[i], in_dims[i])
    a.nsrc, a.M, a.rpb, a.in_dim, a.mid, a.n_mid, a.out_dim = len(xd), M, M, K, mid, n_mid, out_dim
    a.w_in, a.b_in, a.w_mid, a.b_mid = d["w_in"].data_ptr(), d["b_in"].data_ptr(), d["w_mid"].data_ptr(), d["b_mid"].data_ptr()
    a.gn_gamma, a.gn_beta, a.w_out, a.b_out = d["g"].data_ptr(), d["be"].data_ptr(), d["w_out"].data_ptr(), d["b_out"].data_ptr()
    a.gn_eps, a.lrelu_slope = 1e-5, 0.2
    a.out, a.ob, a.orow = out.data_ptr(), 0, out_dim
    if gauss:
        nz = out_dim // 2
        eps = torch.randn(M, nz)
        epsd, z = eps.to(dev), torch.zeros(M, nz, device=dev)
        a.epi, a.eps, a.eb, a.erow, a.z, a.zb, a.zrow = rt.MLP_GAUSS, epsd.data_ptr(), 0, nz, z.data_ptr(), 0, nz
    rt.check(lib.gcpx_mlp(C.byref(a), _stream()), "mlp")
    torch.cuda.synchronize()
    assert_close(out, want, atol=3e-5, rtol=1e-5, name="mlp out")
    if gauss:
        assert_close(z, want[:, :nz] + torch.exp(want[:, nz:]) * eps, atol=5e-5, rtol=1e-5, name="z")


@pytest.mark.parametrize("Mg,Ng,nb", [(32, 512, 6), (64, 512, 6), (128, 512, 6), (256, 512, 2), (64, 128, 1)])
def test_mlp_group_with_gemm(env, Mg, Ng, nb):
    """gcpx_mlp_group_gemm (level_pre_kernel): two Predictors and one batched row GEMM as ONE launch — prior + posterior + the
    split_linear merge of a tree level — equals gcpx_mlp_group followed by gcpx_gemm bit for bit (same workgroup bodies), for every
    GEMM tiling the tree levels use (split-K single tiles, split-K blocks, one-wavefront blocks)."""
    rt, pk, lib, dev = env
    torch.manual_seed(Mg + Ng)
    mid, n_mid, M = 128, 3, Mg
    keep = []

    def predictor(in_dims, out_dim, gauss):
        K = sum(in_dims)
        Ws = dict(w_in=torch.randn(mid, K) / K ** 0.5, b_in=torch.randn(mid) * 0.1, w_mid=torch.randn(n_mid, mid, mid) / mid ** 0.5,
                  b_mid=torch.randn(n_mid, mid) * 0.1, g=torch.rand(n_mid, mid) + 0.5, be=torch.randn(n_mid, mid) * 0.1,
                  w_out=torch.randn(out_dim, mid) / mid ** 0.5, b_out=torch.randn(out_dim) * 0.1)
        d = {k: v.to(dev) for k, v in dict(
            w_in=pk.pack_gemm(Ws["w_in"]), b_in=Ws["b_in"], w_mid=torch.stack([pk.pack_gemm(Ws["w_mid"][i]) for i in range(n_mid)]),
            b_mid=Ws["b_mid"], g=Ws["g"], be=Ws["be"], w_out=pk.pack_gemm(Ws["w_out"]), b_out=Ws["b_out"]).items()}
        xd = [torch.randn(M, dd, device=dev) for dd in in_dims]
        outs = [torch.full((M, out_dim), float("nan"), device=dev) for _ in range(2)]
        zs = [torch.zeros(M, out_dim // 2, device=dev) for _ in range(2)]
        eps = torch.randn(M, out_dim // 2, device=dev)
        args = []
        for o, z in zip(outs, zs):
            a = rt.MlpArgs()
            for i, x in enumerate(xd):
                a.src[i] = _rowsrc(rt, x, 0, in_dims[i], in_dims[i])
            a.nsrc, a.M, a.rpb, a.in_dim, a.mid, a.n_mid, a.out_dim = len(xd), M, M, K, mid, n_mid, out_dim
            a.w_in, a.b_in, a.w_mid, a.b_mid = d["w_in"].data_ptr(), d["b_in"].data_ptr(), d["w_mid"].data_ptr(), d["b_mid"].data_ptr()
            a.gn_gamma, a.gn_beta, a.w_out, a.b_out = d["g"].data_ptr(), d["be"].data_ptr(), d["w_out"].data_ptr(), d["b_out"].data_ptr()
            a.gn_eps, a.lrelu_slope = 1e-5, 0.2
            a.out, a.ob, a.orow = o.data_ptr(), 0, out_dim
            if gauss:
                a.epi, a.eps, a.eb, a.erow, a.z, a.zb, a.zrow = rt.MLP_GAUSS, eps.data_ptr(), 0, out_dim // 2, z.data_ptr(), 0, out_dim // 2
            args.append(a)
        keep.extend([d, xd, eps])
        return args, outs, zs

    pa, pouts, _ = predictor((128, 128), 512, False)
    qa, qouts, qz = predictor((128, 128, 128), 512, True)
    Kg = 1024
    xg = torch.randn(nb, Mg, Kg, device=dev)
    wg = torch.randn(nb, Ng, Kg) / Kg ** 0.5
    wpg = torch.stack([pk.pack_gemm(wg[i]) for i in range(nb)]).contiguous().to(dev)
    bg = torch.randn(nb, Ng, device=dev)
    gouts = [torch.full((nb, Mg, Ng), float("nan"), device=dev) for _ in range(2)]
    gargs = []
    for o in gouts:
        g = rt.GemmArgs()
        g.src[0] = _rowsrc(rt, xg, 0, Kg, Kg)
        g.nsrc, g.M, g.N, g.K, g.rpb = 1, Mg, Ng, Kg, Mg
        g.wpk, g.bias, g.out, g.ob, g.orow = wpg.data_ptr(), bg.data_ptr(), o.data_ptr(), 0, Ng
        if nb > 1:
            g.nbatch, g.z_src_off, g.z_w_off, g.z_bias_off, g.z_out_off = nb, Mg * Kg, Ng * Kg, Ng, Mg * Ng
        gargs.append(g)

    def group(i):
        tab = (rt.MlpArgs * 2)(pa[i], qa[i])
        dims = (C.c_int32 * 8)()
        total = C.c_int32()
        rt.check(lib.gcpx_mlp_group_dims(tab, 2, dims, C.byref(total)), "dims")
        raw = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(dev)
        dd = torch.tensor(list(dims), dtype=torch.int32, device=dev)
        keep.extend([raw, dd, tab])
        return raw, dd, total.value

    raw, dd, total = group(0)
    rt.check(lib.gcpx_mlp_group(raw.data_ptr(), dd.data_ptr(), 2, total, mid, _stream()), "mlp_group")
    rt.check(lib.gcpx_gemm(C.byref(gargs[0]), _stream()), "gemm")
    raw, dd, total = group(1)
    assert lib.gcpx_mlp_group_gemm_supported(C.byref(gargs[1]), total, mid) == 1
    rt.check(lib.gcpx_mlp_group_gemm(raw.data_ptr(), dd.data_ptr(), 2, total, mid, C.byref(gargs[1]), _stream()), "mlp_group_gemm")
    torch.cuda.synchronize()
    want = torch.einsum("bmk,bnk->bmn", xg.cpu().double(), wg.double()) + bg.cpu().double()[:, None]
    assert_close(gouts[1], want.float(), atol=3e-5, rtol=2e-5, name="merge GEMM")
    assert torch.equal(gouts[0], gouts[1]) and torch.equal(pouts[0], pouts[1]) and torch.equal(qouts[0], qouts[1]) and torch.equal(qz[0], qz[1])


def test_loss_pre_and_final_equal_the_single_combine(env):
    """gcpx_loss_pre (KL per sequence + the five terms that need no decoded frame, one launch in front of the decoder) followed by
    gcpx_loss_final == gcpx_kl_gauss + gcpx_loss_combine (base_gcp.py:264-304): every loss value and the total, bit for bit (the same
    device functions, the same reduction orders)."""
    rt, pk, lib, dev = env
    torch.manual_seed(3)
    B, T, N, nz, sd_, na = 5, 20, 31, 32, 2, 2
    q, p_ = torch.randn(B, N, 2 * nz, device=dev) * 0.5, torch.randn(B, N, 2 * nz, device=dev) * 0.5
    end = torch.randint(2, T, (B,), dtype=torch.int64, device=dev)
    seq_len = (end + 1).to(torch.int32)
    pad = (torch.arange(T, device=dev)[None] <= end[:, None]).float()
    t = dict(nll=torch.rand(B, T, device=dev) * 100, len_logits=torch.randn(B, T, device=dev), exist=torch.randn(B, N, device=dev),
             leave=torch.randint(0, 2, (B, N), dtype=torch.int32, device=dev), reg=torch.randn(B, T, sd_, device=dev),
             tgt=torch.randn(B, T, sd_, device=dev), act=torch.randn(B, na, device=dev), acts=torch.randn(B, T - 1, na, device=dev),
             t0=torch.randint(0, T - 1, (B,), dtype=torch.int64, device=dev), cost=torch.randn(B, device=dev), ctgt=torch.randn(B, device=dev))
    outs, kls = [], []
    for mode in ("combine", "pre+final"):
        out, kl = torch.full((16,), float("nan"), device=dev), torch.full((B,), float("nan"), device=dev)
        la = rt.LossArgs()
        la.nll_bt, la.pad_mask, la.kl_b, la.len_logits, la.end_ind = t["nll"].data_ptr(), pad.data_ptr(), kl.data_ptr(), t["len_logits"].data_ptr(), end.data_ptr()
        la.existence, la.leave, la.regressed_state, la.state_target, la.seq_len = t["exist"].data_ptr(), t["leave"].data_ptr(), t["reg"].data_ptr(), t["tgt"].data_ptr(), seq_len.data_ptr()
        la.action_pred, la.action_seq, la.inv_t0, la.cost_pred, la.cost_target = t["act"].data_ptr(), t["acts"].data_ptr(), t["t0"].data_ptr(), t["cost"].data_ptr(), t["ctgt"].data_ptr()
        la.out, la.B, la.T, la.N, la.state_dim, la.n_actions = out.data_ptr(), B, T, N, sd_, na
        la.w_rec, la.w_kl, la.w_len, la.w_exist, la.w_state, la.w_action, la.w_cost, la.total_div = 1.0, 0.5, 1.0, 1.0, 1.0, 2.0, 1.0, 960.0
        klargs = (N, nz, N * 2 * nz, 2 * nz, C.c_float(0.1), None, 0, kl.data_ptr())
        if mode == "combine":
            rt.check(lib.gcpx_kl_gauss(q.data_ptr(), p_.data_ptr(), B, *klargs, _stream()), "kl")
            rt.check(lib.gcpx_loss_combine(C.byref(la), _stream()), "combine")
        else:
            rt.check(lib.gcpx_loss_pre(C.byref(la), q.data_ptr(), p_.data_ptr(), *klargs, _stream()), "pre")
            rt.check(lib.gcpx_loss_final(C.byref(la), _stream()), "final")
        torch.cuda.synchronize()
        outs.append(out[:9].clone()); kls.append(kl)
    assert torch.isfinite(outs[0]).all() and bool((outs[0][[0, 1, 2, 3, 4, 7, 8]] != 0).all())
    assert torch.equal(kls[0], kls[1])
    assert_close(outs[1], outs[0], atol=0, rtol=2e-6, name="loss values")     # (block sums over 256 vs 1024 threads: a different tree)


def _conv_args(rt, srcs, **kw):
    a = rt.ConvArgs()
    a._keep = (srcs, kw)          # device tensors must outlive the launch (the struct only holds raw pointers)
    cin = 0
    for i, (t, Cc, fdiv, sc, sh, act) in enumerate(srcs):
        s = a.src[i]
        s.ptr, s.C, s.frame_div, s.act = t.data_ptr(), Cc, fdiv, act
        s.scale = sc.data_ptr() if sc is not None else None
        s.shift = sh.data_ptr() if sh is not None else None
        cin += Cc
    a.nsrc, a.Cin = len(srcs), cin
    for k, v in kw.items():
        setattr(a, k, v.data_ptr() if hasattr(v, "data_ptr") else v)
    return a


@pytest.mark.parametrize("Hin,c_prev,c_skip,cout,Fr,nodes", [(32, 16, 16, 16, 6, 3), (16, 32, 0, 16, 5, 1), (8, 64, 64, 32, 6, 2),
                                                             (4, 128, 0, 64, 7, 1), (4, 64, 64, 32, 6, 3), (8, 32, 0, 16, 4, 1)])
@pytest.mark.parametrize("split", [False, True, "fold"])
def test_conv3x3_upsample_blocks(env, Hin, c_prev, c_skip, cout, Fr, nodes, split):
    """decoder block: concat(prev, skip broadcast over nodes) -> affine+LReLU -> bilinear x2 -> conv3x3 (+ stats).
    split: the split-f16 forms (csrc/conv3x3_split.hip: wave-autonomous for 16 output channels, workgroup-tiled for 32 / 64), same
    tolerances.  fold: the row-folded form of the 32 -> 16 channel blocks (GCPX_SPLIT_ROWFOLD)."""
    rt, pk, lib, dev = env
    if split == "fold" and not (cout == 16 and c_prev + c_skip == 32 and Hin % 4 == 0):
        pytest.skip("row-folded form: 32 -> 16 channels")
    torch.manual_seed(Hin + cout)
    x = torch.randn(Fr, c_prev, Hin, Hin)
    sc, sh = torch.rand(c_prev) + 0.5, torch.randn(c_prev) * 0.2
    xin = F.leaky_relu(x * sc[None, :, None, None] + sh[None, :, None, None], 0.2)
    srcs_ref = [xin]
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    srcs = [(xd, c_prev, 1, sc.to(dev), sh.to(dev), rt.ACT_LRELU)]
    if c_skip:
        sk = torch.randn(Fr // nodes, c_skip, Hin, Hin)
        srcs_ref.append(sk.repeat_interleave(nodes, 0))
        skd = sk.permute(0, 2, 3, 1).contiguous().to(dev)
        srcs.append((skd, c_skip, nodes, None, None, rt.ACT_NONE))
    cin = c_prev + c_skip
    w, b = torch.randn(cout, cin, 3, 3) / (9 * cin) ** 0.5, torch.randn(cout) * 0.1
    want = F.conv2d(F.interpolate(torch.cat(srcs_ref, 1), scale_factor=2, mode="bilinear", align_corners=False), w, b, padding=1)
    wp, bd = pk.pack_conv3x3(w, 16 if cout == 16 else 32).to(dev), pk.pad_vec(b, cout).to(dev)
    out = torch.full((Fr, 2 * Hin, 2 * Hin, cout), float("nan"), device=dev)
    a = _conv_args(rt, srcs, F=Fr, Hin=Hin, Win=Hin, Hout=2 * Hin, Wout=2 * Hin, Cout=cout, out_pitch=cout, upsample=1,
                   head_mode=rt.HEAD_RAW, wpk=wp, bias=bd, out=out, stats_partial=out)
    if split == "fold":
        ws, e = pk.pack_conv3x3_fold(w)
        ws = ws.to(dev)
        a.wpk_split, a.w_split_log2, a.split_layout = ws.data_ptr(), e, rt.SPLIT_ROWFOLD
    elif split:
        ws, e = pk.pack_conv3x3_split(w) if cout == 16 else pk.pack_conv3x3_split32(w)
        ws = ws.to(dev)
        a.wpk_split, a.w_split_log2 = ws.data_ptr(), e
    G = lib.gcpx_conv3x3_grid(C.byref(a))
    assert G > 0
    st = torch.full((G, 2, cout), float("nan"), device=dev)
    a.stats_partial = st.data_ptr()
    rt.check(lib.gcpx_conv3x3(C.byref(a), _stream()), "conv3x3")
    torch.cuda.synchronize()
    assert_close(out.permute(0, 3, 1, 2), want, atol=2e-5, rtol=1e-5, name="conv3x3 up")
    s = st.sum(0).cpu()
    assert_close(s[0], want.sum((0, 2, 3)), atol=2e-2, rtol=1e-4, name="stats sum")
    assert_close(s[1], (want ** 2).sum((0, 2, 3)), atol=2e-2, rtol=1e-4, name="stats sumsq")


@pytest.mark.parametrize("Hin,c_prev,c_skip,cout,Fr,nodes", [(8, 64, 64, 32, 6, 3), (4, 64, 64, 32, 8, 4), (4, 64, 64, 64, 6, 2)])
def test_conv3x3_up32_skip_channels_convolved_once_per_sequence(env, Hin, c_prev, c_skip, cout, Fr, nodes):
    """gcpx_conv_args.addend: the skip half of a decoder block (the same activations for every node of a sequence) convolved ONCE per
    sequence (second k-steps of the same weight pack, zero bias) and added in the per-node launch's epilogue = the one-launch block over
    the concatenated channels, output and BatchNorm partial sums, within f32 rounding; against F.conv2d as the block test"""
    rt, pk, lib, dev = env
    torch.manual_seed(Hin + cout + nodes)
    x = torch.randn(Fr, c_prev, Hin, Hin)
    sc, sh = torch.rand(c_prev) + 0.5, torch.randn(c_prev) * 0.2
    sk = torch.randn(Fr // nodes, c_skip, Hin, Hin)
    ssc, ssh = torch.rand(c_skip) + 0.5, torch.randn(c_skip) * 0.2
    xin = F.leaky_relu(x * sc[None, :, None, None] + sh[None, :, None, None], 0.2)
    skin = F.leaky_relu(sk * ssc[None, :, None, None] + ssh[None, :, None, None], 0.2)
    cin = c_prev + c_skip
    w, b = torch.randn(cout, cin, 3, 3) / (9 * cin) ** 0.5, torch.randn(cout) * 0.1
    want = F.conv2d(F.interpolate(torch.cat([xin, skin.repeat_interleave(nodes, 0)], 1), scale_factor=2, mode="bilinear", align_corners=False),
                    w, b, padding=1)
    xd, skd = x.permute(0, 2, 3, 1).contiguous().to(dev), sk.permute(0, 2, 3, 1).contiguous().to(dev)
    wp, bd = pk.pack_conv3x3(w, 32).to(dev), pk.pad_vec(b, cout).to(dev)
    ws, e = pk.pack_conv3x3_split32(w)
    ws = ws.to(dev)
    H2 = 2 * Hin
    # one launch over both sources
    out1 = torch.full((Fr, H2, H2, cout), float("nan"), device=dev)
    a1 = _conv_args(rt, [(xd, c_prev, 1, sc.to(dev), sh.to(dev), rt.ACT_LRELU), (skd, c_skip, nodes, ssc.to(dev), ssh.to(dev), rt.ACT_LRELU)],
                    F=Fr, Hin=Hin, Win=Hin, Hout=H2, Wout=H2, Cout=cout, out_pitch=cout, upsample=1, head_mode=rt.HEAD_RAW, wpk=wp, bias=bd,
                    out=out1, stats_partial=out1)
    a1.wpk_split, a1.w_split_log2 = ws.data_ptr(), e
    G = lib.gcpx_conv3x3_grid(C.byref(a1))
    st1 = torch.full((G, 2, cout), float("nan"), device=dev)
    a1.stats_partial = st1.data_ptr()
    rt.check(lib.gcpx_conv3x3(C.byref(a1), _stream()), "block, one launch")
    # the skip half once per sequence, then the nodes' own channels + addend
    add = torch.full((Fr // nodes, H2, H2, cout), float("nan"), device=dev)
    zb = torch.zeros(cout, device=dev)
    a_s = _conv_args(rt, [(skd, c_skip, 1, ssc.to(dev), ssh.to(dev), rt.ACT_LRELU)], F=Fr // nodes, Hin=Hin, Win=Hin, Hout=H2, Wout=H2, Cout=cout,
                     out_pitch=cout, upsample=1, head_mode=rt.HEAD_RAW, wpk=wp, bias=zb, out=add)
    a_s.wpk_split, a_s.w_split_log2 = ws.data_ptr() + (c_prev // 32) * 9 * (cout // 16) * 2048, e
    rt.check(lib.gcpx_conv3x3(C.byref(a_s), _stream()), "skip half")
    out2 = torch.full((Fr, H2, H2, cout), float("nan"), device=dev)
    a2 = _conv_args(rt, [(xd, c_prev, 1, sc.to(dev), sh.to(dev), rt.ACT_LRELU)], F=Fr, Hin=Hin, Win=Hin, Hout=H2, Wout=H2, Cout=cout,
                    out_pitch=cout, upsample=1, head_mode=rt.HEAD_RAW, wpk=wp, bias=bd, out=out2, stats_partial=out2)
    a2.wpk_split, a2.w_split_log2 = ws.data_ptr(), e
    a2.addend, a2.addend_frame_div = add.data_ptr(), nodes
    assert lib.gcpx_conv3x3_grid(C.byref(a2)) == G
    st2 = torch.full((G, 2, cout), float("nan"), device=dev)
    a2.stats_partial = st2.data_ptr()
    rt.check(lib.gcpx_conv3x3(C.byref(a2), _stream()), "nodes' half + addend")
    torch.cuda.synchronize()
    assert_close(out2.permute(0, 3, 1, 2), want, atol=2e-5, rtol=1e-5, name="block with the skip half hoisted vs F.conv2d")
    assert_close(out2, out1, atol=4e-6, rtol=1e-5, name="hoisted vs one launch")
    assert_close(st2.sum(0), st1.sum(0), atol=2e-2, rtol=1e-4, name="BatchNorm partial sums")
    # the exact-f32 kernels do not take an addend: refused, not dropped
    a2.wpk_split = None
    assert lib.gcpx_conv3x3(C.byref(a2), _stream()) != 0


@pytest.mark.parametrize("Hin,Fr,nodes", [(32, 6, 3), (8, 8, 4), (8, 4, 2)])
def test_conv3x3_fold16_block_with_the_skip_half_hoisted(env, Hin, Fr, nodes):
    """GCPX_SPLIT_ROWFOLD16 (conv3x3_up16_fold16_kernel): the 16 + 16 -> 16 channel block as two 16-channel row-folded convs — the skip
    half once per sequence, the node half with that addend — against F.conv2d over the concatenated, upsampled input and against the
    32-channel row-folded kernel (GCPX_SPLIT_ROWFOLD), output and BatchNorm partial sums"""
    rt, pk, lib, dev = env
    torch.manual_seed(Hin + nodes)
    cp = cs = cout = 16
    x = torch.randn(Fr, cp, Hin, Hin)
    sc, sh = torch.rand(cp) + 0.5, torch.randn(cp) * 0.2
    sk = torch.randn(Fr // nodes, cs, Hin, Hin)
    ssc, ssh = torch.rand(cs) + 0.5, torch.randn(cs) * 0.2
    xin = F.leaky_relu(x * sc[None, :, None, None] + sh[None, :, None, None], 0.2)
    skin = F.leaky_relu(sk * ssc[None, :, None, None] + ssh[None, :, None, None], 0.2)
    w, b = torch.randn(cout, cp + cs, 3, 3) / (9 * 32) ** 0.5, torch.randn(cout) * 0.1
    want = F.conv2d(F.interpolate(torch.cat([xin, skin.repeat_interleave(nodes, 0)], 1), scale_factor=2, mode="bilinear", align_corners=False),
                    w, b, padding=1)
    xd, skd = x.permute(0, 2, 3, 1).contiguous().to(dev), sk.permute(0, 2, 3, 1).contiguous().to(dev)
    wp, bd = pk.pack_conv3x3(w, 16).to(dev), pk.pad_vec(b, cout).to(dev)
    H2 = 2 * Hin
    # the 32-channel row-folded kernel, one launch
    wf, ef = pk.pack_conv3x3_fold(w)
    wf = wf.to(dev)
    out1 = torch.full((Fr, H2, H2, cout), float("nan"), device=dev)
    a1 = _conv_args(rt, [(xd, cp, 1, sc.to(dev), sh.to(dev), rt.ACT_LRELU), (skd, cs, nodes, ssc.to(dev), ssh.to(dev), rt.ACT_LRELU)],
                    F=Fr, Hin=Hin, Win=Hin, Hout=H2, Wout=H2, Cout=cout, out_pitch=cout, upsample=1, head_mode=rt.HEAD_RAW, wpk=wp, bias=bd,
                    out=out1, stats_partial=out1)
    a1.wpk_split, a1.w_split_log2, a1.split_layout = wf.data_ptr(), ef, rt.SPLIT_ROWFOLD
    G = lib.gcpx_conv3x3_grid(C.byref(a1))
    st1 = torch.full((G, 2, cout), float("nan"), device=dev)
    a1.stats_partial = st1.data_ptr()
    rt.check(lib.gcpx_conv3x3(C.byref(a1), _stream()), "32-channel row-folded block")
    # the skip half once per sequence, the node half with the addend
    wa, ea = pk.pack_conv3x3_fold16(w, 0)
    wb, eb = pk.pack_conv3x3_fold16(w, 16)
    wa, wb = wa.to(dev), wb.to(dev)
    add = torch.full((Fr // nodes, H2, H2, cout), float("nan"), device=dev)
    a_s = _conv_args(rt, [(skd, cs, 1, ssc.to(dev), ssh.to(dev), rt.ACT_LRELU)], F=Fr // nodes, Hin=Hin, Win=Hin, Hout=H2, Wout=H2, Cout=cout,
                     out_pitch=cout, upsample=1, head_mode=rt.HEAD_RAW, wpk=wp, bias=torch.zeros(cout, device=dev), out=add)
    a_s.wpk_split, a_s.w_split_log2, a_s.split_layout = wb.data_ptr(), eb, rt.SPLIT_ROWFOLD16
    rt.check(lib.gcpx_conv3x3(C.byref(a_s), _stream()), "skip half")
    out2 = torch.full((Fr, H2, H2, cout), float("nan"), device=dev)
    a2 = _conv_args(rt, [(xd, cp, 1, sc.to(dev), sh.to(dev), rt.ACT_LRELU)], F=Fr, Hin=Hin, Win=Hin, Hout=H2, Wout=H2, Cout=cout,
                    out_pitch=cout, upsample=1, head_mode=rt.HEAD_RAW, wpk=wp, bias=bd, out=out2, stats_partial=out2)
    a2.wpk_split, a2.w_split_log2, a2.split_layout = wa.data_ptr(), ea, rt.SPLIT_ROWFOLD16
    a2.addend, a2.addend_frame_div = add.data_ptr(), nodes
    assert lib.gcpx_conv3x3_grid(C.byref(a2)) == G
    st2 = torch.full((G, 2, cout), float("nan"), device=dev)
    a2.stats_partial = st2.data_ptr()
    rt.check(lib.gcpx_conv3x3(C.byref(a2), _stream()), "node half + addend")
    torch.cuda.synchronize()
    assert_close(out2.permute(0, 3, 1, 2), want, atol=2e-5, rtol=1e-5, name="16-channel row-folded block, skip hoisted, vs F.conv2d")
    assert_close(out2, out1, atol=4e-6, rtol=1e-5, name="hoisted vs the 32-channel row-folded launch")
    assert_close(st2.sum(0), st1.sum(0), atol=2e-2, rtol=1e-4, name="BatchNorm partial sums")
    # device-side packs (gcpx_fold_upsample_weights + gcpx_split_pack over conv3x3_fold16_index) = the host packs, bit for bit
    fold = torch.zeros(24 * 16 * 32, device=dev)
    wdev = w.to(dev).contiguous()
    rt.check(lib.gcpx_fold_upsample_weights(wdev.data_ptr(), 16, 32, fold.data_ptr(), _stream()), "fold")
    for cbase, host, eh in ((0, wa, ea), (16, wb, eb)):
        idx = pk.conv3x3_fold16_index(32, cbase).to(dev)
        outp = torch.zeros(2 * idx.numel(), dtype=torch.int16, device=dev)
        lg = torch.zeros(1, dtype=torch.int32, device=dev)
        rt.check(lib.gcpx_split_pack(fold.data_ptr(), idx.data_ptr(), idx.numel(), outp.data_ptr(), lg.data_ptr(), _stream()), "split_pack")
        torch.cuda.synchronize()
        assert int(lg) == eh and torch.equal(outp.view(-1), host.view(-1))


@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("S,Fr", [(32, 3), (64, 2)])
def test_conv3x3_head_dlm(env, S, Fr, split):
    """output head: 16 -> 100 channels; raw parameters (kernel order) + fused mixture mean vs the oracle's formulas.
    split: the split-f16 kernel (csrc/conv3x3_head_split.hip) in place of the exact f32 MFMA kernel, same tolerances."""
    rt, pk, lib, dev = env
    from oracle import gcp_model_oracle as O
    from video_gcp_amd import config
    hp = config("c1")
    torch.manual_seed(S)
    x = torch.randn(Fr, 16, S, S)
    sc, sh = torch.rand(16) + 0.5, torch.randn(16) * 0.2
    xin = F.leaky_relu(x * sc[None, :, None, None] + sh[None, :, None, None], 0.2)
    w, b = torch.randn(100, 16, 3, 3) / 12.0, torch.randn(100) * 0.1
    head = F.conv2d(xin, w, b, padding=1)
    want_img = O.dlm_mean(head, hp)
    perm = pk.dlm_channel_perm(10)
    permt = torch.tensor(perm)
    wp = pk.pack_dlm_head(w, perm).to(dev)
    bk = torch.zeros(len(perm))
    bk[permt >= 0] = b[permt[permt >= 0]]
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    raw = torch.full((Fr, S, S, len(perm)), float("nan"), device=dev)
    img = torch.full((Fr, 3, S, S), float("nan"), device=dev)
    a = _conv_args(rt, [(xd, 16, 1, sc.to(dev), sh.to(dev), rt.ACT_LRELU)], F=Fr, Hin=S, Win=S, Hout=S, Wout=S, Cout=100,
                   out_pitch=len(perm), upsample=0, head_mode=rt.HEAD_DLM_BOTH, wpk=wp, bias=bk.to(dev), out=raw, images=img)
    if split:
        ws, e = pk.pack_conv3x3_split(w, perm)
        ws = ws.to(dev)
        a.wpk_split, a.w_split_log2 = ws.data_ptr(), e
    rt.check(lib.gcpx_conv3x3(C.byref(a), _stream()), "head")
    torch.cuda.synchronize()
    slots = torch.nonzero(permt >= 0)[:, 0]
    inv = torch.empty(100, dtype=torch.long)
    inv[permt[slots]] = slots
    got = raw.cpu().index_select(-1, inv).permute(0, 3, 1, 2)
    assert_close(got, head, atol=2e-5, rtol=1e-5, name="head raw")
    assert_close(img, want_img, atol=1e-5, name="dlm mean")
    # mean-only mode writes the same images
    img2 = torch.full((Fr, 3, S, S), float("nan"), device=dev)
    a.head_mode, a.out, a.images = rt.HEAD_DLM_MEAN, None, img2.data_ptr()
    rt.check(lib.gcpx_conv3x3(C.byref(a), _stream()), "head mean")
    torch.cuda.synchronize()
    assert torch.equal(img2, img)
    if split:
        # the 32x32-tile head (csrc/conv3x3_head32.hip) runs the mean-only mode from its own weight layout: same images within f32
        ws32, e32 = pk.pack_head32_split(w, perm)
        ws32 = ws32.to(dev)
        img32 = torch.full((Fr, 3, S, S), float("nan"), device=dev)
        a.images, a.wpk_split, a.w_split_log2, a.split_layout = img32.data_ptr(), ws32.data_ptr(), e32, rt.SPLIT_HEAD32
        rt.check(lib.gcpx_conv3x3(C.byref(a), _stream()), "head32 mean")
        torch.cuda.synchronize()
        assert_close(img32, want_img, atol=1e-5, name="dlm mean, 32x32 head")
        assert_close(img32, img, atol=4e-6, name="dlm mean, 32x32 head vs 16x16 head")
        a.images, a.wpk_split, a.w_split_log2, a.split_layout = img2.data_ptr(), ws.data_ptr(), e, rt.SPLIT_PLAIN
        # images_rows: frames with a row-map entry are stored at that row of a second array (and of a copy of it) as well; rows no
        # frame maps to stay as they were
        rmap = torch.full((Fr,), -1, dtype=torch.int32)
        rows = [f for f in range(Fr) if f % 2 == 0]
        for i, f in enumerate(reversed(rows)):
            rmap[f] = i
        R = len(rows) + 2
        rows_img = torch.full((2, R, 3, S, S), 7.0, device=dev)
        rmd = rmap.to(dev)
        a.raw_row_map, a.images_rows, a.images_rows_dup = rmd.data_ptr(), rows_img.data_ptr(), rows_img[0].numel()
        img3 = torch.full((Fr, 3, S, S), float("nan"), device=dev)
        a.images = img3.data_ptr()
        rt.check(lib.gcpx_conv3x3(C.byref(a), _stream()), "head mean + rows")
        torch.cuda.synchronize()
        assert torch.equal(img3, img)
        for f in range(Fr):
            if rmap[f] >= 0:
                assert torch.equal(rows_img[0, rmap[f]], img[f]) and torch.equal(rows_img[1, rmap[f]], img[f])
        assert bool((rows_img[:, len(rows):] == 7.0).all())


@pytest.mark.parametrize("layout", ["plain", "head32"])
@pytest.mark.parametrize("case", ["plain", "edges"])
def test_conv3x3_head_fused_likelihood(env, case, layout):
    """GCPX_HEAD_DLM_NLL: the split-f16 head evaluates decoder.nll of the frames matched to a ground-truth frame in its epilogue
    (frame_binding.py:88-99) instead of storing their 100 parameters per pixel for gcpx_dlm_nll.  Checked per frame against (1) the
    stored-parameters path (GCPX_HEAD_DLM_BOTH + gcpx_dlm_nll: same formulas, same fast-math helpers) and (2) the oracle's likelihood
    of a float64 conv.  `edges`: saturated target pixels (x = -1 / +1 take the one-sided branches), log-scales below the -7 clamp and
    bins whose probability vanishes (the log-pdf branch); frame 1 has no matched row and must not be written."""
    rt, pk, lib, dev = env
    from oracle import gcp_model_oracle as O
    from video_gcp_amd import config
    hp = config("c1")
    torch.manual_seed(5)
    S, Fr = 64, 3
    x = torch.randn(Fr, 16, S, S)
    sc, sh = torch.rand(16) + 0.5, torch.randn(16) * 0.2
    xin = F.leaky_relu(x * sc[None, :, None, None] + sh[None, :, None, None], 0.2)
    w, b = torch.randn(100, 16, 3, 3) / 12.0, torch.randn(100) * 0.1
    tgt = torch.rand(2, 3, S, S) * 2 - 1                          # rows of the target tensor: row 0 <- frame 2, row 1 <- frame 0
    if case == "edges":
        b[20:30] -= 4.0                                           # log_scale_r around -4: narrow bins, some vanish
        b[50:60] -= 9.0                                           # log_scale_g below the clamp
        tgt[0, :, :8] = -1.0
        tgt[1, :, 8:16] = 1.0
        tgt[0, 1, 30:34] = 0.9995
    head = F.conv2d(xin.double(), w.double(), b.double(), padding=1)
    want = torch.stack([O.dlm_nll(head[[2]], tgt[[0]].double(), hp)[0].sum(), O.dlm_nll(head[[0]], tgt[[1]].double(), hp)[0].sum()])
    perm = pk.dlm_channel_perm(10)
    permt = torch.tensor(perm)
    wp = pk.pack_dlm_head(w, perm).to(dev)
    ws, e = pk.pack_conv3x3_split(w, perm)
    ws = ws.to(dev)
    bk = torch.zeros(len(perm))
    bk[permt >= 0] = b[permt[permt >= 0]]
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    rows = torch.tensor([1, -1, 0], dtype=torch.int32, device=dev)
    td = tgt.to(dev)
    nit = (S // 4) * (S // 16)
    part = torch.full((nit, 2), float("nan"), device=dev)
    img = torch.full((Fr, 3, S, S), float("nan"), device=dev)
    a = _conv_args(rt, [(xd, 16, 1, sc.to(dev), sh.to(dev), rt.ACT_LRELU)], F=Fr, Hin=S, Win=S, Hout=S, Wout=S, Cout=100,
                   out_pitch=len(perm), upsample=0, head_mode=rt.HEAD_DLM_NLL, wpk=wp, bias=bk.to(dev), out=None, images=img)
    a.raw_row_map, a.wpk_split, a.w_split_log2 = rows.data_ptr(), ws.data_ptr(), e
    a.nll_target, a.nll_partial, a.nll_rows = td.data_ptr(), part.data_ptr(), 2
    if layout == "head32":
        # the 32x32-tile head (csrc/conv3x3_head32.hip): its own weight layout, the same interface and results
        ws32, e32 = pk.pack_head32_split(w, perm)
        ws32 = ws32.to(dev)
        a.wpk_split, a.w_split_log2, a.split_layout = ws32.data_ptr(), e32, rt.SPLIT_HEAD32
    rt.check(lib.gcpx_conv3x3(C.byref(a), _stream()), "head nll")
    got = torch.empty(2, device=dev)
    rt.check(lib.gcpx_reduce_partials(part.data_ptr(), nit, 2, 2, got.data_ptr(), 0, _stream()), "reduce")
    a.wpk_split, a.w_split_log2, a.split_layout = ws.data_ptr(), e, rt.SPLIT_PLAIN
    # the stored-parameters path
    raw = torch.full((2, S, S, len(perm)), float("nan"), device=dev)
    img2 = torch.full((Fr, 3, S, S), float("nan"), device=dev)
    a.head_mode, a.out, a.images = rt.HEAD_DLM_BOTH, raw.data_ptr(), img2.data_ptr()
    rt.check(lib.gcpx_conv3x3(C.byref(a), _stream()), "head both")
    ref = torch.empty(2, device=dev)
    rt.check(lib.gcpx_dlm_nll(raw.data_ptr(), td.data_ptr(), None, ref.data_ptr(), 2, S * S, len(perm), 10, _stream()), "dlm_nll")
    torch.cuda.synchronize()
    assert torch.isfinite(part).all()
    assert_close(img, img2, atol=2e-6 if layout == "plain" else 1e-5, name="mixture mean, likelihood variant vs stored-parameters variant")   # (two instantiations: the compiler contracts them differently)
    assert_close(got, ref, atol=0, rtol=1e-5, name="fused vs stored-parameters likelihood")
    assert_close(got.double().cpu(), want, atol=0, rtol=(1e-4 if case == "edges" else 2e-5), name="fused likelihood vs float64 oracle")
    # without wpk_split the mode is refused (the exact-f32 head keeps the stored-parameters path)
    a.head_mode, a.out, a.wpk_split = rt.HEAD_DLM_NLL, None, None
    assert lib.gcpx_conv3x3(C.byref(a), _stream()) != 0


@pytest.mark.parametrize("layout", ["plain", "head32"])
@pytest.mark.parametrize("case", ["full", "edges"])
def test_conv3x3_head_nll_grad_values(env, case, layout):
    """GCPX_HEAD_DLM_NLL_GRAD (the training forward's head, csrc/conv3x3_head_split.hip, NLL = 2): every stored gradient value of sampled
    frames against autograd of (nll_scale x row weight x) oracle.dlm_nll over a float64 conv of the same inputs, the per-row likelihood
    against the same oracle, and the whole gradient tensor bit for bit across 8 launches.
    `full`: 64 x 64, 2304 frames of which 1408 matched (more than the c2 training forward: both wavefronts of every SIMD busy for the
    whole launch — the regime in which the round-4 store hazard showed), 72 sampled rows.  `edges`: saturated targets (x = -1 / +1),
    log-scales below the -7 clamp (no gradient through the clamp) and bins whose probability vanishes (the bin-centre branch)."""
    rt, pk, lib, dev = env
    from oracle import gcp_model_oracle as O
    from video_gcp_amd import config
    hp = config("c1")
    torch.manual_seed(23)
    S = 64
    Fr, R, n_check = (2304, 1408, 72) if case == "full" else (6, 4, 4)
    x = torch.randn(Fr, S, S, 16)
    sc, sh = torch.rand(16) + 0.5, torch.randn(16) * 0.2
    w, b = torch.randn(100, 16, 3, 3) / 20.0, torch.randn(100) * 0.1
    tgt = torch.rand(R, 3, S, S) * 2 - 1
    if case == "edges":
        b[20:30] -= 4.0                                           # log_scale_r around -4: narrow bins, some vanish
        b[50:60] -= 9.0                                           # log_scale_g below the clamp
        tgt[0, :, :8] = -1.0
        tgt[1, :, 8:16] = 1.0
        tgt[2, 1, 30:34] = 0.9995
    rows = torch.full((Fr,), -1, dtype=torch.int32)
    sel = torch.randperm(Fr)[:R]
    rows[sel] = torch.arange(R, dtype=torch.int32)
    wgt = torch.rand(R) * 0.5 + 0.75
    wgt[R // 2] = 0.0                                             # a padded frame: its row must be written, as zeros
    scale = 1e-3
    perm = pk.dlm_channel_perm(10)
    permt = torch.tensor(perm)
    wp = pk.pack_dlm_head(w, perm).to(dev)
    ws, e = pk.pack_conv3x3_split(w, perm)
    ws = ws.to(dev)
    bk = torch.zeros(len(perm))
    bk[permt >= 0] = b[permt[permt >= 0]]
    xd, rd, td, wd = x.to(dev), rows.to(dev), tgt.to(dev), wgt.to(dev)
    nit = (S // 4) * (S // 16)
    part = torch.full((nit, R), float("nan"), device=dev)
    img = torch.full((Fr, 3, S, S), float("nan"), device=dev)
    grad = torch.full((R, S, S, len(perm)), float("nan"), device=dev)
    a = _conv_args(rt, [(xd, 16, 1, sc.to(dev), sh.to(dev), rt.ACT_LRELU)], F=Fr, Hin=S, Win=S, Hout=S, Wout=S, Cout=100,
                   out_pitch=len(perm), upsample=0, head_mode=rt.HEAD_DLM_NLL_GRAD, wpk=wp, bias=bk.to(dev), out=grad, images=img)
    a.raw_row_map, a.wpk_split, a.w_split_log2 = rd.data_ptr(), ws.data_ptr(), e
    a.nll_target, a.nll_partial, a.nll_rows, a.nll_row_weight, a.nll_scale = td.data_ptr(), part.data_ptr(), R, wd.data_ptr(), scale
    if layout == "head32":
        ws32, e32 = pk.pack_head32_split(w, perm)
        ws32 = ws32.to(dev)
        a.wpk_split, a.w_split_log2, a.split_layout = ws32.data_ptr(), e32, rt.SPLIT_HEAD32
    rt.check(lib.gcpx_conv3x3(C.byref(a), _stream()), "head nll + gradient")
    torch.cuda.synchronize()
    first, first_part = grad.clone(), part.clone()
    for rep in range(7):                                          # run to run: the same bits in all 8 launches
        grad.fill_(float("nan"))
        rt.check(lib.gcpx_conv3x3(C.byref(a), _stream()), "head nll + gradient")
        torch.cuda.synchronize()
        assert torch.equal(grad, first), f"launch {rep + 2} differs from the first in {int((grad != first).sum())} values"
        assert torch.equal(part, first_part)
    assert torch.isfinite(first).all() and torch.isfinite(img).all()
    assert bool((first[..., 100:] == 0).all())                    # the 12 empty slots of the 112-slot layout
    assert bool((first[R // 2] == 0).all())                       # zero row weight
    # sampled rows against autograd in float64
    check = torch.randperm(R)[:n_check].tolist() if case == "full" else list(range(R))
    frame_of = {int(r): f for f, r in enumerate(rows.tolist()) if r >= 0}
    slots = torch.nonzero(permt >= 0)[:, 0]
    inv = torch.empty(100, dtype=torch.long)
    inv[permt[slots]] = slots
    nll_got = first_part.sum(0).double().cpu()
    worst = 0.0
    for r in check:
        f = frame_of[r]
        xin = F.leaky_relu(x[f].permute(2, 0, 1)[None].double() * sc[None, :, None, None].double() + sh[None, :, None, None].double(), 0.2)
        head = F.conv2d(xin, w.double(), b.double(), padding=1).requires_grad_(True)
        nll = O.dlm_nll(head, tgt[[r]].double(), hp).sum()
        (g,) = torch.autograd.grad(nll * (scale * float(wgt[r])), head)
        got = first[r].cpu().index_select(-1, inv).permute(2, 0, 1).double()
        ref = g[0]
        # f32 arithmetic on hardware exp / rcp against float64.  The bin probability is a difference of two exponentials ~0.8 % / scale
        # apart (in the reference: of two sigmoids): one rounding of each is ~1e-5 of the difference, and d / d mean, d / d log_scale
        # divide by it — measured worst case 1.9e-5 of the row's largest gradient (tools/head_grad_probe.py; the round-4 kernel, which
        # formed s (1 - s) as s - s^2, carried 1.5e-4).  A stale or dropped register (the round-4 store hazard) is an error of the
        # size of the value.
        tol = 5e-5 * float(ref.abs().max()) + 2e-4 * ref.abs()
        err = (got - ref).abs()
        worst = max(worst, float((err / (tol + 1e-30)).max()))
        assert bool((err <= tol).all()), (r, float(err.max()), float(ref.abs().max()), float((err / tol).max()))
        assert abs(float(nll_got[r]) - float(nll.detach())) <= (1e-4 if case == "edges" else 2e-5) * abs(float(nll.detach())), (r, float(nll_got[r]), float(nll.detach()))
    print(f"head nll gradient [{case}]: worst error / tolerance over {len(check)} rows = {worst:.3f}")


@pytest.mark.parametrize("case", ["unit", "tiny", "large", "outlier", "zero", "matched_rows"])
def test_conv3x3_head_split_error_vs_float64(env, case):
    """The split-f16 head against a float64 conv of the same f32 inputs, next to the exact f32 MFMA kernel: its error is of the same
    size (three exact f16 x f16 partial products per f32 product, a per-item power-of-two scale) whatever the magnitude of the
    activations — 1e-3, 300, one 3e4 outlier pixel, all zero — and with the matched-rows map of the training forward."""
    rt, pk, lib, dev = env
    torch.manual_seed(11)
    S, Fr = 64, 3
    amp = {"unit": 1.0, "tiny": 1e-3, "large": 300.0, "outlier": 1.0, "zero": 0.0, "matched_rows": 1.0}[case]
    x = torch.randn(Fr, 16, S, S) * amp
    if case == "outlier":
        x[1, 3, 17, 40] = 3e4
    sc, sh = torch.rand(16) + 0.5, torch.randn(16) * 0.2 * amp
    w, b = torch.randn(100, 16, 3, 3) / 12.0, torch.randn(100) * 0.1
    xin = F.leaky_relu(x * sc[None, :, None, None] + sh[None, :, None, None], 0.2)
    ref = F.conv2d(xin.double(), w.double(), b.double(), padding=1)
    perm = pk.dlm_channel_perm(10)
    permt = torch.tensor(perm)
    wp = pk.pack_dlm_head(w, perm).to(dev)
    ws, e = pk.pack_conv3x3_split(w, perm)
    ws = ws.to(dev)
    bk = torch.zeros(len(perm))
    bk[permt >= 0] = b[permt[permt >= 0]]
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    slots = torch.nonzero(permt >= 0)[:, 0]
    inv = torch.empty(100, dtype=torch.long)
    inv[permt[slots]] = slots
    rows = torch.tensor([1, -1, 0], dtype=torch.int32, device=dev) if case == "matched_rows" else None
    n_rows = 2 if rows is not None else Fr
    err, imgs = {}, {}
    for name in ("f32", "split"):
        raw = torch.full((n_rows, S, S, len(perm)), float("nan"), device=dev)
        img = torch.full((Fr, 3, S, S), float("nan"), device=dev)
        a = _conv_args(rt, [(xd, 16, 1, sc.to(dev), sh.to(dev), rt.ACT_LRELU)], F=Fr, Hin=S, Win=S, Hout=S, Wout=S, Cout=100,
                       out_pitch=len(perm), upsample=0, head_mode=rt.HEAD_DLM_BOTH, wpk=wp, bias=bk.to(dev), out=raw, images=img)
        if rows is not None:
            a.raw_row_map = rows.data_ptr()
        if name == "split":
            a.wpk_split, a.w_split_log2 = ws.data_ptr(), e
        rt.check(lib.gcpx_conv3x3(C.byref(a), _stream()), name)
        torch.cuda.synchronize()
        got = raw.cpu().index_select(-1, inv).permute(0, 3, 1, 2).double()
        want = ref if rows is None else ref[[2, 0]]                  # row 0 <- frame 2, row 1 <- frame 0, frame 1 not stored
        assert torch.isfinite(got).all() and torch.isfinite(img).all()
        err[name] = (got - want).abs()
        imgs[name] = img.cpu()
    scale = float(ref.abs().max()) + 1e-30
    # same order as the exact kernel (measured: 0.7x its rms error on unit data), never more than a few f32 roundings of the result
    assert float(err["split"].pow(2).mean().sqrt()) <= 1.5 * float(err["f32"].pow(2).mean().sqrt()) + 1e-12 * scale
    assert float(err["split"].max()) <= 2.0 * float(err["f32"].max()) + 4e-7 * scale
    if case in ("unit", "tiny", "zero", "matched_rows"):
        assert_close(imgs["split"], imgs["f32"], atol=5e-6, name="mixture mean, split vs exact")


@pytest.mark.parametrize("shape,perm10,amp", [((100, 16, 3, 3), True, 1.0), ((16, 32, 3, 3), False, 1e-3), ((16, 32, 3, 3), False, 0.0)])
def test_split_pack_on_device_equals_host_pack(env, shape, perm10, amp):
    """gcpx_split_pack (gather from the flat parameter vector + power-of-two scale + two f16 pieces, one launch) writes the f16 values
    packing.pack_conv3x3_split computes on the host in float64 (compared as numbers: the sign of a zero piece is not kept)."""
    rt, pk, lib, dev = env
    torch.manual_seed(2)
    w = torch.randn(*shape) * amp
    perm = pk.dlm_channel_perm(10) if perm10 else None
    off = 24
    theta = torch.cat([torch.randn(off) * 100, w.reshape(-1), torch.randn(8) * 100]).to(dev)
    idx = pk.conv3x3_split_index(shape, off, perm).to(dev)
    out = torch.full((2 * idx.numel(),), -1, dtype=torch.int16, device=dev)
    e = torch.full((1,), 99, dtype=torch.int32, device=dev)
    rt.check(lib.gcpx_split_pack(theta.data_ptr(), idx.data_ptr(), idx.numel(), out.data_ptr(), e.data_ptr(), _stream()), "split_pack")
    torch.cuda.synchronize()
    want, we = pk.pack_conv3x3_split(w, perm)
    assert int(e) == we
    assert torch.equal(out.cpu().view(torch.float16).view(want.shape), want.view(torch.float16))


@pytest.mark.parametrize("case", ["unit", "chunk_scales", "outlier", "zero_chunk"])
def test_conv3x3_up16_split_error_vs_float64(env, case):
    """The split-f16 16-channel decoder block against float64, next to the exact f32 kernel; the two 16-channel chunks of the input
    differ in magnitude by 1e4 either way (the running power-of-two scale), carry an outlier, or are all zero."""
    rt, pk, lib, dev = env
    torch.manual_seed(5)
    Hin, Fr, nodes = 32, 4, 2
    amp_prev, amp_skip = {"unit": (1, 1), "chunk_scales": (1e-2, 1e2), "outlier": (1, 1), "zero_chunk": (0, 1)}[case]
    res = {}
    for order in (0, 1):                                   # small chunk first / large chunk first
        a_p, a_s = (amp_prev, amp_skip) if order == 0 else (amp_skip, amp_prev)
        x = torch.randn(Fr, 16, Hin, Hin) * a_p
        sk = torch.randn(Fr // nodes, 16, Hin, Hin) * a_s
        if case == "outlier":
            sk[0, 5, 9, 9] = 2e4
        sc, sh = torch.rand(16) + 0.5, torch.randn(16) * 0.2 * a_p
        xin = F.leaky_relu(x * sc[None, :, None, None] + sh[None, :, None, None], 0.2)
        w, b = torch.randn(16, 32, 3, 3) / (9 * 32) ** 0.5, torch.randn(16) * 0.1
        up = F.interpolate(torch.cat([xin, sk.repeat_interleave(nodes, 0)], 1).double(), scale_factor=2, mode="bilinear", align_corners=False)
        ref = F.conv2d(up, w.double(), b.double(), padding=1)
        xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
        skd = sk.permute(0, 2, 3, 1).contiguous().to(dev)
        srcs = [(xd, 16, 1, sc.to(dev), sh.to(dev), rt.ACT_LRELU), (skd, 16, nodes, None, None, rt.ACT_NONE)]
        wp, bd = pk.pack_conv3x3(w, 16).to(dev), pk.pad_vec(b, 16).to(dev)
        ws, e = pk.pack_conv3x3_split(w)
        ws = ws.to(dev)
        err = {}
        for name in ("f32", "split"):
            out = torch.full((Fr, 2 * Hin, 2 * Hin, 16), float("nan"), device=dev)
            a = _conv_args(rt, srcs, F=Fr, Hin=Hin, Win=Hin, Hout=2 * Hin, Wout=2 * Hin, Cout=16, out_pitch=16, upsample=1,
                           head_mode=rt.HEAD_RAW, wpk=wp, bias=bd, out=out)
            if name == "split":
                a.wpk_split, a.w_split_log2 = ws.data_ptr(), e
            rt.check(lib.gcpx_conv3x3(C.byref(a), _stream()), name)
            torch.cuda.synchronize()
            got = out.cpu().permute(0, 3, 1, 2).double()
            assert torch.isfinite(got).all()
            err[name] = (got - ref).abs()
        scale = float(ref.abs().max())
        assert float(err["split"].pow(2).mean().sqrt()) <= 1.5 * float(err["f32"].pow(2).mean().sqrt()) + 1e-12 * scale
        assert float(err["split"].max()) <= 2.0 * float(err["f32"].max()) + 4e-7 * scale


@pytest.mark.parametrize("case", ["unit", "tiny", "large", "outlier", "zero", "zero_skip"])
@pytest.mark.parametrize("Hin,c_prev,c_skip", [(32, 16, 16), (16, 32, 0), (8, 16, 16)])
def test_conv3x3_up16_fold_error_vs_float64(env, case, Hin, c_prev, c_skip):
    """The row-folded split-f16 32 -> 16 channel decoder blocks (additional_conv_layer / pyramid-0 shapes, and the smallest
    image the kernel takes: every item carries a border correction) against float64, next to the exact f32 kernel: the folded weights are f32
    roundings of float64 sums, the border rows subtract the -W0 / -W2 correction — the error stays of the size of the exact
    kernel's whatever the magnitude of the data."""
    rt, pk, lib, dev = env
    torch.manual_seed(13)
    Fr, nodes = 6, 3
    amp = {"unit": 1.0, "tiny": 1e-3, "large": 300.0, "outlier": 1.0, "zero": 0.0, "zero_skip": 1.0}[case]
    x = torch.randn(Fr, c_prev, Hin, Hin) * amp
    if case == "outlier":
        x[1, 3, Hin // 2, 1] = 2e4
    sc, sh = torch.rand(c_prev) + 0.5, torch.randn(c_prev) * 0.2 * amp
    srcs_ref = [F.leaky_relu(x * sc[None, :, None, None] + sh[None, :, None, None], 0.2)]
    srcs = [(x.permute(0, 2, 3, 1).contiguous().to(dev), c_prev, 1, sc.to(dev), sh.to(dev), rt.ACT_LRELU)]
    if c_skip:
        sk = torch.randn(Fr // nodes, c_skip, Hin, Hin) * (0.0 if case == "zero_skip" else amp)
        srcs_ref.append(sk.repeat_interleave(nodes, 0))
        srcs.append((sk.permute(0, 2, 3, 1).contiguous().to(dev), c_skip, nodes, None, None, rt.ACT_NONE))
    w, b = torch.randn(16, 32, 3, 3) / (9 * 32) ** 0.5, torch.randn(16) * 0.1
    up = F.interpolate(torch.cat(srcs_ref, 1).double(), scale_factor=2, mode="bilinear", align_corners=False)
    ref = F.conv2d(up, w.double(), b.double(), padding=1)
    wp, bd = pk.pack_conv3x3(w, 16).to(dev), pk.pad_vec(b, 16).to(dev)
    ws, e = pk.pack_conv3x3_fold(w)
    ws = ws.to(dev)
    err = {}
    for name in ("f32", "fold"):
        out = torch.full((Fr, 2 * Hin, 2 * Hin, 16), float("nan"), device=dev)
        a = _conv_args(rt, srcs, F=Fr, Hin=Hin, Win=Hin, Hout=2 * Hin, Wout=2 * Hin, Cout=16, out_pitch=16, upsample=1,
                       head_mode=rt.HEAD_RAW, wpk=wp, bias=bd, out=out)
        if name == "fold":
            a.wpk_split, a.w_split_log2, a.split_layout = ws.data_ptr(), e, rt.SPLIT_ROWFOLD
        rt.check(lib.gcpx_conv3x3(C.byref(a), _stream()), name)
        torch.cuda.synchronize()
        got = out.cpu().permute(0, 3, 1, 2).double()
        assert torch.isfinite(got).all()
        err[name] = (got - ref).abs()
    scale = float(ref.abs().max())
    # the fold adds one f32 rounding per folded weight and the border correction one more: 2x the exact kernel's rms, 3x its max
    assert float(err["fold"].pow(2).mean().sqrt()) <= 2.0 * float(err["f32"].pow(2).mean().sqrt()) + 1e-12 * scale
    assert float(err["fold"].max()) <= 3.0 * float(err["f32"].max()) + 6e-7 * scale


def test_fold_pack_on_device_equals_host_pack(env):
    """gcpx_fold_upsample_weights + gcpx_split_pack over packing.conv3x3_fold_index write the pieces packing.pack_conv3x3_fold computes"""
    rt, pk, lib, dev = env
    torch.manual_seed(4)
    w = torch.randn(16, 32, 3, 3) * 0.3
    wd = w.to(dev)
    scratch = torch.full((24, 16, 32), float("nan"), device=dev)
    rt.check(lib.gcpx_fold_upsample_weights(wd.data_ptr(), 16, 32, scratch.data_ptr(), _stream()), "fold")
    torch.cuda.synchronize()
    assert torch.equal(scratch.cpu(), pk.fold_up_weights(w))
    idx = pk.conv3x3_fold_index().to(dev)
    out = torch.full((2 * idx.numel(),), -1, dtype=torch.int16, device=dev)
    e = torch.full((1,), 99, dtype=torch.int32, device=dev)
    rt.check(lib.gcpx_split_pack(scratch.data_ptr(), idx.data_ptr(), idx.numel(), out.data_ptr(), e.data_ptr(), _stream()), "split_pack")
    torch.cuda.synchronize()
    want, we = pk.pack_conv3x3_fold(w)
    assert int(e) == we
    assert torch.equal(out.cpu().view(torch.float16).view(want.shape), want.view(torch.float16))


@pytest.mark.parametrize("case", ["unit", "chunk_scales", "outlier", "zero_chunk"])
@pytest.mark.parametrize("Hin,c_prev,c_skip,cout", [(8, 64, 64, 32), (4, 128, 0, 64), (4, 64, 64, 32)])
def test_conv3x3_up32_split_error_vs_float64(env, case, Hin, c_prev, c_skip, cout):
    """The split-f16 32 / 64-channel decoder blocks (pyramid-1 / pyramid-2 shapes and the 8x8x4-frame tile with two channel tiles)
    against float64, next to the exact f32 kernel; the 32-channel chunks of the input differ in magnitude by 1e4 either way (the running
    power-of-two scale is per (tile, chunk)), carry an outlier, or are all zero."""
    rt, pk, lib, dev = env
    torch.manual_seed(7)
    Fr, nodes = 6, 3
    cin = c_prev + c_skip
    amp_a, amp_b = {"unit": (1, 1), "chunk_scales": (1e-2, 1e2), "outlier": (1, 1), "zero_chunk": (0, 1)}[case]
    for order in (0, 1):
        a_lo, a_hi = (amp_a, amp_b) if order == 0 else (amp_b, amp_a)
        camp = torch.where(torch.arange(cin) < cin // 2, torch.tensor(float(a_lo)), torch.tensor(float(a_hi)))
        x = torch.randn(Fr, c_prev, Hin, Hin) * camp[None, :c_prev, None, None]
        if case == "outlier":
            x[2, 7, 1, 2] = 2e4
        sc, sh = torch.rand(c_prev) + 0.5, torch.randn(c_prev) * 0.2 * camp[:c_prev]
        srcs_ref = [F.leaky_relu(x * sc[None, :, None, None] + sh[None, :, None, None], 0.2)]
        xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
        srcs = [(xd, c_prev, 1, sc.to(dev), sh.to(dev), rt.ACT_LRELU)]
        if c_skip:
            sk = torch.randn(Fr // nodes, c_skip, Hin, Hin) * camp[None, c_prev:, None, None]
            srcs_ref.append(sk.repeat_interleave(nodes, 0))
            srcs.append((sk.permute(0, 2, 3, 1).contiguous().to(dev), c_skip, nodes, None, None, rt.ACT_NONE))
        w, b = torch.randn(cout, cin, 3, 3) / (9 * cin) ** 0.5, torch.randn(cout) * 0.1
        up = F.interpolate(torch.cat(srcs_ref, 1).double(), scale_factor=2, mode="bilinear", align_corners=False)
        ref = F.conv2d(up, w.double(), b.double(), padding=1)
        wp, bd = pk.pack_conv3x3(w, 32).to(dev), pk.pad_vec(b, cout).to(dev)
        ws, e = pk.pack_conv3x3_split32(w)
        ws = ws.to(dev)
        err = {}
        for name in ("f32", "split"):
            out = torch.full((Fr, 2 * Hin, 2 * Hin, cout), float("nan"), device=dev)
            a = _conv_args(rt, srcs, F=Fr, Hin=Hin, Win=Hin, Hout=2 * Hin, Wout=2 * Hin, Cout=cout, out_pitch=cout, upsample=1,
                           head_mode=rt.HEAD_RAW, wpk=wp, bias=bd, out=out)
            if name == "split":
                a.wpk_split, a.w_split_log2 = ws.data_ptr(), e
            rt.check(lib.gcpx_conv3x3(C.byref(a), _stream()), name)
            torch.cuda.synchronize()
            got = out.cpu().permute(0, 3, 1, 2).double()
            assert torch.isfinite(got).all()
            err[name] = (got - ref).abs()
        scale = float(ref.abs().max())
        assert float(err["split"].pow(2).mean().sqrt()) <= 1.5 * float(err["f32"].pow(2).mean().sqrt()) + 1e-12 * scale
        assert float(err["split"].max()) <= 2.0 * float(err["f32"].max()) + 4e-7 * scale


@pytest.mark.parametrize("Hin,cin,cout,Fr", [(32, 16, 32, 5), (16, 32, 64, 3), (8, 64, 128, 9), (16, 16, 32, 2), (8, 32, 64, 7), (8, 32, 64, 60),
                                             (16, 16, 32, 63)])
@pytest.mark.parametrize("split", [False, True])
def test_conv4x4s2(env, Hin, cin, cout, Fr, split):
    """encoder block vs torch; split: the split-f16 form (csrc/conv_enc_split.hip), same tolerances"""
    rt, pk, lib, dev = env
    torch.manual_seed(Hin + cin)
    x = torch.randn(Fr, cin, Hin, Hin)
    sc, sh = torch.rand(cin) + 0.5, torch.randn(cin) * 0.2
    w, b = torch.randn(cout, cin, 4, 4) / (16 * cin) ** 0.5, torch.randn(cout) * 0.1
    want = F.conv2d(F.leaky_relu(x * sc[None, :, None, None] + sh[None, :, None, None], 0.2), w, b, stride=2, padding=1)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    out = torch.full((Fr, Hin // 2, Hin // 2, cout), float("nan"), device=dev)
    G = lib.gcpx_conv4x4s2_grid()
    st = torch.full((G, 2, cout), float("nan"), device=dev)
    a = _conv_args(rt, [(xd, cin, 1, sc.to(dev), sh.to(dev), rt.ACT_LRELU)], F=Fr, Hin=Hin, Win=Hin, Hout=Hin // 2, Wout=Hin // 2,
                   Cout=cout, out_pitch=cout, wpk=pk.pack_conv4x4(w).to(dev), bias=b.to(dev), out=out, stats_partial=st)
    if split:
        ws, e = pk.pack_conv4x4_split(w)
        ws = ws.to(dev)
        a.wpk_split, a.w_split_log2 = ws.data_ptr(), e
    rt.check(lib.gcpx_conv4x4s2(C.byref(a), _stream()), "conv4x4s2")
    torch.cuda.synchronize()
    assert_close(out.permute(0, 3, 1, 2), want, atol=2e-5, rtol=1e-5, name="conv4x4s2")
    assert_close(st.sum(0)[0], want.sum((0, 2, 3)), atol=1e-2, rtol=1e-4, name="stats")
    assert_close(st.sum(0)[1], (want ** 2).sum((0, 2, 3)), atol=1e-2, rtol=1e-4, name="stats sumsq")


@pytest.mark.parametrize("case", ["unit", "tiny", "large", "outlier", "zero"])
@pytest.mark.parametrize("Hin,cin,cout,Fr", [(32, 16, 32, 3), (16, 32, 64, 9), (8, 64, 128, 19)])
def test_conv4x4s2_split_error_vs_float64(env, case, Hin, cin, cout, Fr):
    """The split-f16 encoder blocks against float64, next to the exact f32 kernel (one power-of-two scale per staged block of
    frames): data of magnitude 1e-3, 300, with an outlier, all zero; frame counts that leave the last block partly empty."""
    rt, pk, lib, dev = env
    torch.manual_seed(17)
    amp = {"unit": 1.0, "tiny": 1e-3, "large": 300.0, "outlier": 1.0, "zero": 0.0}[case]
    x = torch.randn(Fr, cin, Hin, Hin) * amp
    if case == "outlier":
        x[1, 3, 2, 5] = 3e4
    sc, sh = torch.rand(cin) + 0.5, torch.randn(cin) * 0.2 * amp
    w, b = torch.randn(cout, cin, 4, 4) / (16 * cin) ** 0.5, torch.randn(cout) * 0.1
    ref = F.conv2d(F.leaky_relu(x * sc[None, :, None, None] + sh[None, :, None, None], 0.2).double(), w.double(), b.double(), stride=2, padding=1)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wp, bd = pk.pack_conv4x4(w).to(dev), b.to(dev)
    ws, e = pk.pack_conv4x4_split(w)
    ws = ws.to(dev)
    err = {}
    for name in ("f32", "split"):
        out = torch.full((Fr, Hin // 2, Hin // 2, cout), float("nan"), device=dev)
        a = _conv_args(rt, [(xd, cin, 1, sc.to(dev), sh.to(dev), rt.ACT_LRELU)], F=Fr, Hin=Hin, Win=Hin, Hout=Hin // 2, Wout=Hin // 2,
                       Cout=cout, out_pitch=cout, wpk=wp, bias=bd, out=out)
        if name == "split":
            a.wpk_split, a.w_split_log2 = ws.data_ptr(), e
        rt.check(lib.gcpx_conv4x4s2(C.byref(a), _stream()), name)
        torch.cuda.synchronize()
        got = out.cpu().permute(0, 3, 1, 2).double()
        assert torch.isfinite(got).all()
        err[name] = (got - ref).abs()
    scale = float(ref.abs().max())
    assert float(err["split"].pow(2).mean().sqrt()) <= 1.5 * float(err["f32"].pow(2).mean().sqrt()) + 1e-12 * scale
    assert float(err["split"].max()) <= 2.0 * float(err["f32"].max()) + 4e-7 * scale


@pytest.mark.parametrize("S,Fr", [(32, 5), (64, 3)])
def test_conv4x4s2_image(env, S, Fr):
    rt, pk, lib, dev = env
    torch.manual_seed(S)
    x = torch.rand(Fr, 3, S, S) * 2 - 1
    w, b = torch.randn(16, 3, 4, 4) / 7.0, torch.randn(16) * 0.1
    want = F.leaky_relu(F.conv2d(x, w, b, stride=2, padding=1), 0.2)
    out = torch.full((Fr, S // 2, S // 2, 16), float("nan"), device=dev)
    xd, wp, bd = x.to(dev), pk.pack_conv4x4_image(w).to(dev), b.to(dev)
    rt.check(lib.gcpx_conv4x4s2_image(xd.data_ptr(), wp.data_ptr(), bd.data_ptr(), out.data_ptr(), Fr, S, S, 16, rt.ACT_LRELU,
                                      _stream()), "conv image")
    torch.cuda.synchronize()
    assert_close(out.permute(0, 3, 1, 2), want, atol=1e-5, rtol=1e-5, name="conv image")


def test_balanced_binding_bit_exact(env):
    rt, pk, lib, dev = env
    from oracle import tree_index_oracle as TI
    import numpy as np
    for L, T in [(3, 7), (5, 20), (7, 80), (8, 200)]:
        ends = list(range(0, min(T, 2 ** L - 1)))
        B = len(ends)
        N = 2 ** L - 1
        end = torch.tensor(ends, dtype=torch.long, device=dev)
        node_t, leave = torch.zeros(B, N, dtype=torch.int32, device=dev), torch.zeros(B, N, dtype=torch.int32, device=dev)
        f2n, et = torch.zeros(B, T, dtype=torch.int32, device=dev), torch.zeros(B * N, dtype=torch.int32, device=dev)
        sl, kept = torch.zeros(B, dtype=torch.int32, device=dev), torch.zeros(B, T, dtype=torch.int32, device=dev)
        n2r = torch.zeros(B, N, dtype=torch.int32, device=dev)
        rt.check(lib.gcpx_balanced_binding(end.data_ptr(), B, L, T, node_t.data_ptr(), leave.data_ptr(), f2n.data_ptr(),
                                           et.data_ptr(), sl.data_ptr(), n2r.data_ptr(), _stream()), "binding")
        rt.check(lib.gcpx_compact_index(leave.data_ptr(), B, N, T, kept.data_ptr(), _stream()), "compact")
        torch.cuda.synchronize()
        perm = TI.bf2df_perm(L)
        ts_bf = TI.balanced_timesteps_bf(ends, L, T)
        md = TI.balanced_match_dist(ends, L, T)
        want_t = np.zeros_like(ts_bf)
        want_t[:, perm] = ts_bf
        assert np.array_equal(node_t.cpu().numpy(), want_t)
        assert np.array_equal(leave.cpu().numpy().astype(bool), TI.leave_mask_df(ends, L, T))
        want_f2n = perm[TI.matched_node_index(md)]
        assert np.array_equal(f2n.cpu().numpy(), want_f2n)
        assert np.array_equal(sl.cpu().numpy(), np.array(ends) + 1)
        lv = TI.leave_mask_df(ends, L, T)
        assert np.array_equal(n2r.cpu().numpy(), np.where(lv, np.arange(B)[:, None] * T + want_t, -1))
        k = kept.cpu().numpy()
        for b, e in enumerate(ends):
            assert np.array_equal(k[b, :e + 1], np.nonzero(TI.leave_mask_df([e], L, T)[0])[0])
            assert np.all(k[b, e + 1:] == -1)
        # posterior gather rows, bf order per level
        off = 0
        for l in range(L):
            n = 2 ** l
            blk = et[B * (n - 1):B * (n - 1) + B * n].cpu().numpy().reshape(B, n)
            assert np.array_equal(blk, np.arange(B)[:, None] * T + np.clip(ts_bf[:, off:off + n], 0, T - 1))
            off += n


def test_gemm_batched_projections(env):
    """blockIdx.z batching: the 2*n_lstm_layers split_linear projections (tree_lstm.py:46-47) in one launch."""
    rt, pk, lib, dev = env
    torch.manual_seed(11)
    M, H, nb = 24, 64, 6
    h1, h2 = torch.randn(M, nb * H), torch.randn(M, nb * H)
    Ws, bs = torch.randn(nb, H, 2 * H) / (2 * H) ** 0.5, torch.randn(nb, H)
    want = torch.cat([F.linear(torch.cat([h1[:, z * H:(z + 1) * H], h2[:, z * H:(z + 1) * H]], 1), Ws[z], bs[z]) for z in range(nb)], 1)
    h1d, h2d = h1.to(dev), h2.to(dev)
    wp = torch.stack([pk.pack_gemm(Ws[z]) for z in range(nb)]).contiguous().to(dev)
    bd = bs.contiguous().to(dev)
    out = torch.full((M, nb * H), float("nan"), device=dev)
    a = rt.GemmArgs()
    a.src[0] = _rowsrc(rt, h1d, 0, nb * H, H)
    a.src[1] = _rowsrc(rt, h2d, 0, nb * H, H)
    a.nsrc, a.M, a.N, a.K, a.rpb = 2, M, H, 2 * H, M
    a.wpk, a.bias, a.out, a.ob, a.orow = wp.data_ptr(), bd.data_ptr(), out.data_ptr(), 0, nb * H
    a.nbatch, a.z_src_off, a.z_w_off, a.z_bias_off, a.z_out_off = nb, H, wp[0].numel(), H, H
    rt.check(lib.gcpx_gemm(C.byref(a), _stream()), "gemm batched")
    torch.cuda.synchronize()
    assert_close(out, want, atol=2e-5, rtol=1e-5, name="batched gemm")


@pytest.mark.parametrize("mid,in_dims,out_dim,M,rpb", [(128, (256, 128), 512, 37, 37), (128, (256,), 512, 64, 8), (32, (128, 128), 80, 16, 16),
                                                      (128, (128,), 1, 5, 5)])
def test_mlp_bwd_fused(env, mid, in_dims, out_dim, M, rpb):
    """gcpx_mlp (forward, with `save`) + gcpx_mlp_bwd against torch autograd over the same Predictor: every du_l, the GroupNorm
    parameter sums and the input gradients (ragged last row block, rows addressed as (b, j))."""
    rt, pk, lib, dev = env
    torch.manual_seed(mid + out_dim + M)
    n_mid, K = 3, sum(in_dims)
    xs = [torch.randn(M, d, requires_grad=True) for d in in_dims]
    Ws = dict(w_in=torch.randn(mid, K) / K ** 0.5, b_in=torch.randn(mid) * 0.1,
              w_mid=torch.randn(n_mid, mid, mid) / mid ** 0.5, b_mid=torch.randn(n_mid, mid) * 0.1,
              g=(torch.rand(n_mid, mid) + 0.5).requires_grad_(), be=(torch.randn(n_mid, mid) * 0.1).requires_grad_(),
              w_out=torch.randn(out_dim, mid) / mid ** 0.5, b_out=torch.randn(out_dim) * 0.1)
    want = _predictor_ref(torch.cat(xs, 1), Ws, n_mid)
    dout = torch.randn(M, out_dim)
    grads = torch.autograd.grad(want, xs + [Ws["g"], Ws["be"]], dout)
    out_pad = (out_dim + 15) // 16 * 16
    d = {k: v.detach().to(dev) for k, v in dict(
        w_in=pk.pack_gemm(Ws["w_in"]), b_in=Ws["b_in"], w_mid=torch.stack([pk.pack_gemm(Ws["w_mid"][i]) for i in range(n_mid)]),
        b_mid=Ws["b_mid"], g=Ws["g"], be=Ws["be"], w_out=pk.pack_gemm(Ws["w_out"]), b_out=pk.pad_vec(Ws["b_out"], out_pad)).items()}
    xd = [x.detach().to(dev) for x in xs]
    out = torch.zeros(M, out_dim, device=dev)
    save = torch.zeros(1 + 2 * n_mid, M, mid, device=dev)
    a = rt.MlpArgs()
    for i, x in enumerate(xd):
        a.src[i] = _rowsrc(rt, x, 0, in_dims[i], in_dims[i])
    a.nsrc, a.M, a.rpb, a.in_dim, a.mid, a.n_mid, a.out_dim = len(xd), M, M, K, mid, n_mid, out_dim
    a.w_in, a.b_in, a.w_mid, a.b_mid = d["w_in"].data_ptr(), d["b_in"].data_ptr(), d["w_mid"].data_ptr(), d["b_mid"].data_ptr()
    a.gn_gamma, a.gn_beta, a.w_out, a.b_out = d["g"].data_ptr(), d["be"].data_ptr(), d["w_out"].data_ptr(), d["b_out"].data_ptr()
    a.gn_eps, a.lrelu_slope = 1e-5, 0.2
    a.out, a.ob, a.orow, a.save = out.data_ptr(), 0, out_dim, save.data_ptr()
    rt.check(lib.gcpx_mlp(C.byref(a), _stream()), "mlp")

    w_out_pad = torch.cat([Ws["w_out"], torch.zeros(out_pad - out_dim, mid)], 0)
    wT_out = pk.pack_gemm(w_out_pad.t().contiguous()).to(dev)
    wT_mid = [pk.pack_gemm(Ws["w_mid"][l].t().contiguous()).to(dev) for l in range(n_mid)]
    doutd = torch.zeros(M, out_pad, device=dev)
    doutd[:, :out_dim] = dout.to(dev)
    nb = lib.gcpx_mlp_bwd_blocks(M)
    du = [torch.full((M, mid), float("nan"), device=dev) for _ in range(n_mid + 1)]
    parts = [torch.full((nb, 2, mid), float("nan"), device=dev) for _ in range(n_mid)]
    nbatch = M // rpb
    b = rt.MlpBwdArgs()
    b.dout, b.save, b.wT_out, b.ldo = doutd.data_ptr(), save.data_ptr(), wT_out.data_ptr(), out_pad
    b.M, b.rpb, b.mid, b.n_mid, b.out_pad, b.ndx = M, rpb, mid, n_mid, out_pad, len(in_dims)
    b.gn_eps, b.lrelu_slope = 1e-5, 0.2
    b.du[0] = du[0].data_ptr()
    for l in range(n_mid):
        b.wT_mid[l], b.gn_gamma[l], b.gn_beta[l] = wT_mid[l].data_ptr(), d["g"][l].data_ptr(), d["be"][l].data_ptr()
        b.du[1 + l], b.gn_partial[l] = du[1 + l].data_ptr(), parts[l].data_ptr()
    dxs, wTs, c0 = [], [], 0
    for i, w in enumerate(in_dims):
        wTs.append(pk.pack_gemm(Ws["w_in"][:, c0:c0 + w].t().contiguous()).to(dev))
        # rows (b, j) land in a padded [nbatch][rpb + 3][w + 16] buffer: exercises ob / orow
        dxs.append(torch.full((nbatch, rpb + 3, w + 16), float("nan"), device=dev))
        b.dx[i].wT, b.dx[i].out, b.dx[i].ob, b.dx[i].orow, b.dx[i].width = wTs[i].data_ptr(), dxs[i].data_ptr(), (rpb + 3) * (w + 16), w + 16, w
        c0 += w
    rt.check(lib.gcpx_mlp_bwd(C.byref(b), _stream()), "mlp_bwd")
    torch.cuda.synchronize()
    for i, w in enumerate(in_dims):
        got = dxs[i][:, :rpb, :w].reshape(M, w)
        assert_close(got, grads[i], atol=2e-4, rtol=1e-4, name=f"dx{i}")
        assert torch.isnan(dxs[i][:, rpb:, :]).all() and torch.isnan(dxs[i][:, :, w:]).all(), "wrote outside the addressed rows"
    for l in range(n_mid):
        assert_close(parts[l][:, 0].sum(0), grads[len(in_dims)][l], atol=5e-4, rtol=1e-4, name=f"dgamma{l}")
        assert_close(parts[l][:, 1].sum(0), grads[len(in_dims) + 1][l], atol=5e-4, rtol=1e-4, name=f"dbeta{l}")
    assert not any(torch.isnan(t).any() for t in du)


def test_wgrad_group_matches_single_launches(env):
    """gcpx_wgrad_group over a table of direct and split problems == the same problems launched one by one (bit-exact: same kernel
    body, same accumulation order)."""
    rt, pk, lib, dev = env
    torch.manual_seed(7)
    shapes = [(100, 128, 256, 1), (64, 512, 128, 1), (300, 64, 64, 2), (33, 16, 128, 1), (256, 128, 1024, 4)]    # R, N, K, nsplit
    keep, probs = [], []
    for (R, N, K, ns) in shapes:
        dy, x = torch.randn(R, N, device=dev), torch.randn(R, K, device=dev)
        outs = [torch.zeros(ns, N, K, device=dev) if ns > 1 else torch.zeros(N, K, device=dev) for _ in range(2)]
        args = []
        for o in outs:
            a = rt.WgradArgs()
            a.dy, a.x, a.ldy, a.R, a.N, a.n_valid, a.K, a.mode = dy.data_ptr(), x.data_ptr(), N, R, N, N, K, rt.WG_ROWS
            a.sb, a.sr, a.rpb = R * K, K, R
            a.out, a.ldw, a.accumulate, a.partial, a.nsplit = o.data_ptr(), K, (0 if ns > 1 else 1), (1 if ns > 1 else 0), ns
            args.append(a)
        keep += [dy, x, outs]
        probs.append((args, outs, dy, x, ns))
    for row_split in (0, -1):
        by_variant = {}
        v, nb = C.c_int32(), C.c_int32()
        for args, outs, *_ in probs:
            for o in outs:
                o.zero_()
            rt.check(lib.gcpx_wgrad(C.byref(args[0]), _stream()), "wgrad")          # note: its own heuristic variant
            rt.check(lib.gcpx_wgrad_classify(C.byref(args[1]), row_split, C.byref(v), C.byref(nb)), "classify")
            by_variant.setdefault(v.value, []).append((args[1], nb.value))
        for var, items in by_variant.items():
            tab = (rt.WgradArgs * len(items))(*[it[0] for it in items])
            raw = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(dev)
            starts, tot = [], 0
            for it in items:
                starts.append(tot)
                tot += it[1]
            bst = torch.tensor(starts, dtype=torch.int32, device=dev)
            rt.check(lib.gcpx_wgrad_group(raw.data_ptr(), bst.data_ptr(), len(items), tot, var, _stream()), "wgrad_group")
        torch.cuda.synchronize()
        for args, outs, dy, x, ns in probs:
            want = dy.t() @ x
            got = outs[1].sum(0) if ns > 1 else outs[1]
            assert_close(got, want, atol=2e-3, rtol=1e-4, name="grouped dW")
            one = outs[0].sum(0) if ns > 1 else outs[0]
            assert_close(got, one, atol=2e-4, rtol=1e-5, name="grouped vs single")


@pytest.mark.parametrize("R,N,K", [(1024, 256, 128), (300, 128, 256), (512, 512, 384)])
@pytest.mark.parametrize("case", ["plain", "row_scales", "tiny", "strided_batch", "grouped"])
def test_wgrad_rows_split_error_vs_float64(env, R, N, K, case):
    """Weight gradients of the tree's Linear / LSTM layers on the split-f16 kernel (csrc/wgrad_rows_split.hip: gcpx_wgrad_args.split_f16,
    whole 128 x 128 blocks of dW, >= 256 plain rows) against float64, next to the exact f32 kernel on the same descriptor: dW, the
    fused bias gradient into two destinations, accumulation into existing gradients.  Row blocks whose magnitudes differ by six
    orders (the running scale drops, the sums are rescaled), gradients of order 1e-6, a row count that is no multiple of 64;
    'strided_batch': rows addressed as (batch element, node) with strides for both operands and three batched problems (the
    split_linear projections); 'grouped': through gcpx_wgrad_classify (variant 4) + gcpx_wgrad_group."""
    rt, pk, lib, dev = env
    torch.manual_seed(R + N + K)
    nb = 3 if case == "strided_batch" else 1
    rpb = 4 if case == "strided_batch" else R
    B = R // rpb
    ldy, sr = nb * N + 32, nb * K + 16                          # row pitches wider than the problem (other columns belong to other problems)
    dy_sb, sb = rpb * ldy + 64, rpb * sr + 128                  # batch elements further apart than their rows
    dyb = torch.randn(B, dy_sb // ldy + 1, ldy, device=dev)
    xb = torch.randn(B, sb // sr + 1, sr, device=dev)
    dyf, xf = dyb.reshape(B, -1)[:, :dy_sb].contiguous(), xb.reshape(B, -1)[:, :sb].contiguous()      # [B][dy_sb], [B][sb] flat batch slabs
    if case == "row_scales":
        dyf *= torch.logspace(-3, 3, B, device=dev)[:, None]
        xf *= torch.logspace(2, -2, B, device=dev)[:, None]
    elif case == "tiny":
        dyf *= 1e-6

    def rows(flat, pitch, bstride, width, z):                   # [R][width] view of batch problem z
        idx = (torch.arange(B, device=dev)[:, None] * bstride + torch.arange(rpb, device=dev)[None] * pitch).reshape(-1)
        return flat.reshape(-1)[(idx[:, None] + z * width + torch.arange(width, device=dev)[None])]

    ref = [rows(dyf, ldy, dy_sb, N, z).double().t() @ rows(xf, sr, sb, K, z).double() for z in range(nb)]
    refb = [rows(dyf, ldy, dy_sb, N, z).double().sum(0) for z in range(nb)]
    ldw, k_off = K + 64, 32
    err, berr = {}, {}
    init = 0.0 if case == "tiny" else 1.0                       # existing gradient the launch accumulates onto
    for name, split in (("f32", 0), ("split", 1)):
        out = torch.full((nb, N, ldw), init, device=dev)
        db, db2 = torch.full((nb, N), init, device=dev), torch.full((N,), init, device=dev)
        a = rt.WgradArgs()
        a.dy, a.x, a.ldy, a.R, a.N, a.n_valid, a.K, a.mode = dyf.data_ptr(), xf.data_ptr(), ldy, R, N, N, K, rt.WG_ROWS
        a.sb, a.sr, a.rpb, a.dy_sb, a.dy_rpb = sb, sr, rpb, dy_sb, rpb
        a.out, a.ldw, a.k_off, a.accumulate, a.partial, a.nsplit = out.data_ptr(), ldw, k_off, 1, 0, 1
        a.dbias, a.dbias2 = db.data_ptr(), (db2.data_ptr() if nb == 1 else None)
        if nb > 1:
            a.nbatch, a.z_dy_off, a.z_x_off, a.z_out_off, a.z_bias_off = nb, N, K, N * ldw, N
        a.split_f16 = split
        if case == "grouped" and split:
            v, nblk = C.c_int32(), C.c_int32()
            rt.check(lib.gcpx_wgrad_classify(C.byref(a), 0, C.byref(v), C.byref(nblk)), "classify")
            assert v.value == 4 and nblk.value == (N // 128) * (K // 128) * nb
            tab = (rt.WgradArgs * 1)(a)
            raw = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(dev)
            bst = torch.zeros(1, dtype=torch.int32, device=dev)
            rt.check(lib.gcpx_wgrad_group(raw.data_ptr(), bst.data_ptr(), 1, nblk.value, 4, _stream()), "wgrad_group")
        else:
            rt.check(lib.gcpx_wgrad(C.byref(a), _stream()), name)
        torch.cuda.synchronize()
        assert torch.isfinite(out).all()
        assert bool((out[:, :, :k_off] == init).all()) and bool((out[:, :, k_off + K:] == init).all())       # neighbours untouched
        err[name] = torch.stack([(out[z, :, k_off:k_off + K].double() - init - ref[z]).abs() for z in range(nb)])
        berr[name] = torch.stack([(db[z].double() - init - refb[z]).abs() for z in range(nb)])
        if nb == 1:
            assert torch.equal(db2, db[0])
    scale = max(float(r.abs().max()) for r in ref) + 1e-300
    e32, esp = float(err["f32"].max()) / scale, float(err["split"].max()) / scale
    r32, rsp = float(err["f32"].pow(2).mean().sqrt()) / scale, float(err["split"].pow(2).mean().sqrt()) / scale
    # (the outputs accumulate onto `init`: one more f32 rounding of max(init, |dW|) in both kernels)
    floor = 1.2e-7 * max(init, scale) / scale
    assert esp <= 2.0 * e32 + 4e-7 + floor, (esp, e32)
    assert rsp <= 1.5 * r32 + 1e-7 + floor, (rsp, r32)
    bscale = max(float(r.abs().max()) for r in refb) + 1e-300
    assert float(berr["split"].max()) / bscale <= 2.0 * float(berr["f32"].max()) / bscale + 4e-7 + 1.2e-7 * max(init, bscale) / bscale


@pytest.mark.parametrize("S,cin,cout,Fr,use_frames", [(32, 112, 16, 9, True), (32, 112, 16, 9, False), (64, 48, 16, 5, True),
                                                      (32, 16, 32, 6, None), (64, 16, 32, 3, None)])
@pytest.mark.parametrize("split", [False, True, "tiny"])
def test_conv3x3_plain_row_maps(env, S, cin, cout, Fr, use_frames, split):
    """Plain 3x3 conv (the data gradients of the training step): frames read source rows through src_row_map (negative: zeros);
    with the inverse map from gcpx_index_inverse the wave-autonomous kernel walks the rows, without it the tiled kernel runs —
    both against F.conv2d on the gathered rows.  A padded row that no frame reads must not leak into any frame.
    split: the split-f16 form of the wave-autonomous kernel (conv3x3_wave_split_kernel); "tiny": on data of magnitude 1e-6 with
    chunks that differ 100-fold (loss gradients: the per-item scale has to follow the chunks), error relative to the result."""
    rt, pk, lib, dev = env
    if split and use_frames is False:
        pytest.skip("without the inverse map the tiled exact kernel runs")
    torch.manual_seed(S + cin + Fr)
    w = torch.randn(cout, cin, 3, 3) / (9 * cin) ** 0.5
    wp, bd = pk.pack_conv3x3(w, 16).to(dev), torch.zeros(64, device=dev)
    if use_frames is None:                                 # every frame has its own row
        R, fmap = Fr, None
    else:                                                  # frames 1, 4, ... have no row; one extra row (index 2) belongs to nobody
        rows = [r for r in range(Fr) if r % 3 != 1]
        R = len(rows) + 1
        fmap = torch.full((Fr,), -1, dtype=torch.int32)
        free = [r for r in range(R) if r != 2]
        for f, r in zip(rows, free):
            fmap[f] = r
    x = torch.randn(R, cin, S, S)
    if split == "tiny":
        x *= 1e-6 * torch.logspace(0, 2, cin)[None, :, None, None]
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    out = torch.full((Fr, S, S, cout), float("nan"), device=dev)
    a = _conv_args(rt, [(xd, cin, 1, None, None, rt.ACT_NONE)], F=Fr, Hin=S, Win=S, Hout=S, Wout=S, Cout=cout, out_pitch=cout,
                   upsample=0, head_mode=rt.HEAD_RAW, wpk=wp, bias=bd, out=out)
    if split:
        ws, e = pk.pack_conv3x3_split(w)
        ws = ws.to(dev)
        a.wpk_split, a.w_split_log2 = ws.data_ptr(), e
    if fmap is not None:
        fd = fmap.to(dev)
        a.src_row_map = fd.data_ptr()
        if use_frames:
            inv = torch.full((R,), 12345, dtype=torch.int32, device=dev)
            rt.check(lib.gcpx_index_inverse(fd.data_ptr(), Fr, inv.data_ptr(), R, _stream()), "index_inverse")
            torch.cuda.synchronize()
            want_inv = torch.full((R,), -1, dtype=torch.int32)
            for f in range(Fr):
                if fmap[f] >= 0:
                    want_inv[fmap[f]] = f
            assert torch.equal(inv.cpu(), want_inv)
            a.src_row_frames, a.n_src_rows = inv.data_ptr(), R
    rt.check(lib.gcpx_conv3x3(C.byref(a), _stream()), "conv3x3 plain")
    torch.cuda.synchronize()
    full = F.conv2d(x, w, None, padding=1)
    want = torch.zeros(Fr, cout, S, S)
    for f in range(Fr):
        r = f if fmap is None else int(fmap[f])
        if r >= 0:
            want[f] = full[r]
    scale = float(want.abs().max())
    assert_close(out.permute(0, 3, 1, 2), want, atol=3e-5 * min(1.0, scale), rtol=1e-5, name="plain conv3x3")


@pytest.mark.parametrize("with_rows", [True, False])
def test_conv3x3_data_gradient_with_activation_backward(env, with_rows):
    """gcpx_conv_args.bwd_r: the head's data gradient (112 -> 16 channels, split-f16 wave-autonomous kernel) with the next step of the
    backward pass in its epilogue — g = conv * LeakyReLU'(scale r + shift) stored, per-workgroup sums of g and g * x_hat left for
    gcpx_bn_bwd_finalize — against the plain conv followed by gcpx_act_bwd (what the training step ran before); with and without
    the matched-row maps (frames without a row: zeros, no contribution)."""
    rt, pk, lib, dev = env
    torch.manual_seed(5)
    S, cin, cout, Fr = 32, 112, 16, 7
    w = torch.randn(cout, cin, 3, 3) / (9 * cin) ** 0.5
    wp, bd = pk.pack_conv3x3(w, 16).to(dev), torch.zeros(64, device=dev)
    ws, e = pk.pack_conv3x3_split(w)
    ws = ws.to(dev)
    if with_rows:
        rows = [r for r in range(Fr) if r % 3 != 1]
        fmap = torch.full((Fr,), -1, dtype=torch.int32)
        for i, f in enumerate(rows):
            fmap[f] = i
        R = len(rows) + 1                                        # one row nobody reads
    else:
        fmap, R = None, Fr
    x = (torch.randn(R, S, S, cin) * 1e-4).to(dev)               # loss gradients: small
    r = torch.randn(Fr, S, S, cout, device=dev) * 2 + 0.5
    scale, shift = torch.randn(cout, device=dev), torch.randn(cout, device=dev) * 0.3
    mean, rstd = torch.randn(cout, device=dev) * 0.2 + 0.5, torch.rand(cout, device=dev) + 0.3

    def run(fused):
        out = torch.full((Fr, S, S, cout), float("nan"), device=dev)
        a = _conv_args(rt, [(x, cin, 1, None, None, rt.ACT_NONE)], F=Fr, Hin=S, Win=S, Hout=S, Wout=S, Cout=cout, out_pitch=cout,
                       upsample=0, head_mode=rt.HEAD_RAW, wpk=wp, bias=bd, out=out)
        a.wpk_split, a.w_split_log2 = ws.data_ptr(), e
        keep = []
        if fmap is not None:
            fd = fmap.to(dev)
            inv = torch.full((R,), -1, dtype=torch.int32, device=dev)
            rt.check(lib.gcpx_index_inverse(fd.data_ptr(), Fr, inv.data_ptr(), R, _stream()), "index_inverse")
            a.src_row_map, a.src_row_frames, a.n_src_rows = fd.data_ptr(), inv.data_ptr(), R
            keep += [fd, inv]
        st = None
        if fused:
            nb = lib.gcpx_conv3x3_grid(C.byref(a))
            st = torch.full((nb, 2, cout), float("nan"), device=dev)
            a.bwd_r, a.bwd_scale, a.bwd_shift, a.bwd_mean, a.bwd_rstd = r.data_ptr(), scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), rstd.data_ptr()
            a.stats_partial = st.data_ptr()
        rt.check(lib.gcpx_conv3x3(C.byref(a), _stream()), "conv3x3 plain")
        torch.cuda.synchronize()
        return out, st

    plain, _ = run(False)
    dy = torch.empty_like(plain)
    nb = lib.gcpx_act_bwd_blocks()
    st_ref = torch.zeros(nb, 2, cout, device=dev)
    b = rt.ActBwdArgs()
    b.da, b.r, b.scale, b.shift, b.mean, b.rstd = plain.data_ptr(), r.data_ptr(), scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), rstd.data_ptr()
    b.dy, b.stats_partial, b.ldc, b.c_off, b.up, b.fsum, b.act = dy.data_ptr(), st_ref.data_ptr(), cout, 0, 0, 1, rt.ACT_LRELU
    b.F, b.H, b.W, b.C = Fr, S, S, cout
    rt.check(lib.gcpx_act_bwd(C.byref(b), _stream()), "act_bwd")
    torch.cuda.synchronize()
    got, st = run(True)
    assert torch.isfinite(st).all()
    assert torch.equal(got, dy)                                  # same products, same mask
    want, have = st_ref.double().sum(0), st.double().sum(0)
    assert_close(have, want, atol=2e-6 * float(want.abs().max()), rtol=1e-5, name="BatchNorm backward sums")


def test_aux_sample_indices_uniform_and_gauss(env):
    """gcpx_aux_sample_indices (four uniform numbers per sequence) and its _gauss twin (four standard-normal numbers, u = Phi(n): they
    share the generator launch of the latent noise and the kernel sits inside the forward's graph) against the host formulas of
    synthetic.aux_indices (InverseModel.sample_offsets / CostModel._general_cost index ranges, inverse_mdl.py:84-104, cost_mdl.py:105-107)."""
    rt, pk, lib, dev = env
    from video_gcp_amd.synthetic import aux_indices
    torch.manual_seed(21)
    B, T = 257, 80
    end = torch.randint(2, T, (B,), dtype=torch.int64)
    n = torch.randn(4, B)
    n[0, :4] = torch.tensor([-9.0, 9.0, 0.0, 5.5])                 # Phi at the ends of the float range
    u_from_n = (0.5 * torch.erfc(-n.double() * 0.70710678118654752440))
    u = torch.rand(4, B)
    outs = {}
    for name, fn, src, uu in (("uniform", lib.gcpx_aux_sample_indices, u, u.double()), ("gauss", lib.gcpx_aux_sample_indices_gauss, n, u_from_n)):
        want = aux_indices(end, uu, 1)
        d = [torch.full((B,), -7, dtype=torch.int64, device=dev) for _ in range(4)]
        end_d, src_d = end.to(dev), src.to(dev)
        rt.check(fn(end_d.data_ptr(), src_d.data_ptr(), B, 1, *[t.data_ptr() for t in d], _stream()), name)
        torch.cuda.synchronize()
        for t, k in zip(d, ("inv_t0", "inv_t1", "cost_start_idx", "cost_end_idx")):
            assert torch.equal(t.cpu(), want[k]), (name, k)
        t0, t1, cs, ce = [t.cpu() for t in d]
        assert bool(((t0 >= 0) & (t1 > t0) & (t1 <= end) & (cs >= 0) & (ce > cs) & (ce <= end)).all())


@pytest.mark.parametrize("tiles", ["4x16", "2x32"])
@pytest.mark.parametrize("Cout,Cin,S,Fr", [(100, 16, 64, 5), (16, 32, 64, 5), (16, 64, 64, 3), (32, 64, 32, 7), (64, 128, 16, 9),
                                           (64, 128, 8, 12), (32, 64, 8, 10)])
@pytest.mark.parametrize("case", ["plain", "frame_scales", "tiny", "zero_frames"])
def test_wgrad_conv3x3_split_error_vs_float64(env, Cout, Cin, S, Fr, case, tiles, monkeypatch):
    """Weight gradient of the decoder's 3x3 convs on the split-f16 kernel (csrc/wgrad_conv_split.hip: both operands split in the
    kernel, quad transposes on the way into LDS, running power-of-two scales per workgroup) against float64, next to the exact f32
    MFMA kernel.  Frames whose magnitudes differ by six orders (the running scale drops, the sums are rescaled), gradients of
    order 1e-6, all-zero frames; every tile shape (W = 64 / 32 / 16 / 8) and channel configuration of the decoder."""
    if tiles == "2x32":          # 64-pixel tiles as 2 rows x 32 columns (the form before round 6; images of 32+ columns)
        monkeypatch.setenv("GCPX_WS_TW32", "1")
    rt, pk, lib, dev = env
    torch.manual_seed(Cout + Cin + S)
    N16 = (Cout + 15) // 16 * 16
    dy = torch.randn(Fr, S, S, N16, device=dev)
    dy[..., Cout:] = 0.0
    u = torch.randn(Fr, S, S, Cin, device=dev)
    if case == "frame_scales":
        dy *= torch.logspace(-3, 3, Fr, device=dev)[:, None, None, None]
        u *= torch.logspace(2, -2, Fr, device=dev)[:, None, None, None]
    elif case == "tiny":
        dy *= 1e-6
    elif case == "zero_frames":
        dy[::2] = 0.0
        u[1::3] = 0.0
    up = F.pad(u.double(), (0, 0, 1, 1, 1, 1))
    ref = torch.stack([torch.einsum("fhwn,fhwc->nc", dy.double(), up[:, ky:ky + S, kx:kx + S, :]) for ky in range(3) for kx in range(3)], 1)
    ref = ref.reshape(N16, 9 * Cin)                    # [n][tap*Cin + ci]
    ych = Cin // 32 if (Cin % 32 == 0 and N16 != 112) else Cin // 16
    grid = max(1, min(96 // ych, Fr * max(1, S * S // 64)))
    err = {}
    for name, fn in (("f32", lib.gcpx_wgrad_conv3x3), ("split", lib.gcpx_wgrad_conv3x3_split)):
        part = torch.full((grid, N16, 9 * Cin), float("nan"), device=dev)
        rt.check(fn(dy.data_ptr(), N16, u.data_ptr(), Fr, S, S, Cin, Cout, part.data_ptr(), grid, _stream()), name)
        torch.cuda.synchronize()
        assert torch.isfinite(part).all()
        err[name] = (part.double().sum(0) - ref).abs()
    scale = float(ref.abs().max()) + 1e-300
    e32, esp = float(err["f32"].max()) / scale, float(err["split"].max()) / scale
    r32, rsp = float(err["f32"].pow(2).mean().sqrt()) / scale, float(err["split"].pow(2).mean().sqrt()) / scale
    assert esp <= 2.0 * e32 + 4e-7, (esp, e32)
    assert rsp <= 1.5 * r32 + 1e-7, (rsp, r32)


@pytest.mark.parametrize("S,Fr,with_add", [(64, 5, False), (64, 3, True), (32, 7, True), (128, 2, False)])
def test_wgrad_image4x4s2(env, S, Fr, with_add):
    """First encoder layer's weight + bias gradient in one launch (csrc/wgrad_image.hip: LeakyReLU backward, image patches, both
    sums) against torch's conv2d weight gradient in float64; image borders (zero padding), every band of output rows, the optional
    skip-gradient addend, fewer workgroups than items (several items per persistent workgroup) and more (idle workgroups write zeros)."""
    rt, pk, lib, dev = env
    torch.manual_seed(S + Fr)
    x = torch.randn(Fr, 3, S, S, device=dev)
    H2 = S // 2
    da, r = torch.randn(Fr, H2, H2, 16, device=dev), torch.randn(Fr, H2, H2, 16, device=dev)
    add = torch.randn(Fr, H2, H2, 16, device=dev) if with_add else None
    g = (da + (add if with_add else 0.0)).double() * torch.where(r > 0, 1.0, 0.2).double()
    gn = g.permute(0, 3, 1, 2).contiguous()
    dW = torch.nn.grad.conv2d_weight(x.double(), (16, 3, 4, 4), gn, stride=2, padding=1).reshape(16, 48)
    db = gn.sum((0, 2, 3))
    nitems = Fr * (H2 // 8)
    for grid in (max(1, nitems // 3), nitems + 5):
        part = torch.full((grid, 16 * 48 + 16), float("nan"), device=dev)
        rt.check(lib.gcpx_wgrad_image4x4s2(da.data_ptr(), add.data_ptr() if with_add else None, r.data_ptr(), x.data_ptr(), Fr, S,
                                           part.data_ptr(), grid, _stream()), "wgrad_image")
        torch.cuda.synchronize()
        assert torch.isfinite(part).all()
        tot = part.double().sum(0)
        assert float((tot[:768].view(16, 48) - dW).abs().max()) <= 2e-5 * float(dW.abs().max())
        assert float((tot[768:] - db).abs().max()) <= 2e-5 * float(db.abs().max()) + 1e-4


@pytest.mark.parametrize("Hin,c_prev,c_skip,Fr,nodes,up", [(32, 16, 16, 6, 3, 1), (8, 64, 64, 6, 2, 1), (4, 128, 0, 5, 1, 1), (16, 16, 0, 4, 1, 0)])
def test_conv_stage(env, Hin, c_prev, c_skip, Fr, nodes, up):
    """gcpx_conv_stage: the (bilinear x2, align_corners=False) concat of the normalised + activated sources that the decoder's weight
    gradients read — 2 x 2 output pixels per low-resolution pixel, clamped borders — against torch; with a frame map (skipped frames
    are written as zeros)."""
    rt, pk, lib, dev = env
    torch.manual_seed(Hin + c_prev)
    x = torch.randn(Fr, c_prev, Hin, Hin)
    sc, sh = torch.rand(c_prev) + 0.5, torch.randn(c_prev) * 0.2
    ref = [F.leaky_relu(x * sc[None, :, None, None] + sh[None, :, None, None], 0.2)]
    srcs = [(x.permute(0, 2, 3, 1).contiguous().to(dev), c_prev, 1, sc.to(dev), sh.to(dev), rt.ACT_LRELU)]
    if c_skip:
        sk = torch.randn(Fr // nodes, c_skip, Hin, Hin)
        ref.append(sk.repeat_interleave(nodes, 0))
        srcs.append((sk.permute(0, 2, 3, 1).contiguous().to(dev), c_skip, nodes, None, None, rt.ACT_NONE))
    cin = c_prev + c_skip
    want = torch.cat(ref, 1)
    if up:
        want = F.interpolate(want, scale_factor=2, mode="bilinear", align_corners=False)
    Ho = Hin * (2 if up else 1)
    out = torch.full((Fr, Ho, Ho, cin), float("nan"), device=dev)
    a = _conv_args(rt, srcs, F=Fr, Hin=Hin, Win=Hin, Hout=Ho, Wout=Ho, Cout=cin, out_pitch=cin, upsample=up, out=out)
    rt.check(lib.gcpx_conv_stage(C.byref(a), _stream()), "conv_stage")
    torch.cuda.synchronize()
    assert_close(out.permute(0, 3, 1, 2), want, atol=2e-6, rtol=1e-6, name="conv_stage")
    if not c_skip:
        fmap = torch.arange(Fr, dtype=torch.int32)
        fmap[1] = -1
        fmap[Fr - 1] = 0
        fm = fmap.to(dev)
        out.fill_(float("nan"))
        a.src_row_map = fm.data_ptr()
        rt.check(lib.gcpx_conv_stage(C.byref(a), _stream()), "conv_stage")
        torch.cuda.synchronize()
        w2 = want[fmap.clamp(min=0).long()].clone()
        w2[1] = 0.0
        assert_close(out.permute(0, 3, 1, 2), w2, atol=2e-6, rtol=1e-6, name="conv_stage mapped")


@pytest.mark.parametrize("tiles", ["4x16", "2x32"])
@pytest.mark.parametrize("Hin,c_prev,c_skip,Fr,nodes", [(32, 16, 16, 6, 3), (16, 32, 0, 5, 1), (8, 32, 0, 7, 1), (4, 16, 16, 8, 2)])
@pytest.mark.parametrize("case", ["plain", "tiny"])
def test_wgrad_conv3x3_split_up_reads_the_blocks_sources(env, Hin, c_prev, c_skip, Fr, nodes, case, tiles, monkeypatch):
    """gcpx_wgrad_conv3x3_split_up: weight gradient of a 16-output-channel upsampling block that forms its operand (bilinear x2 of the
    concatenated, normalised + activated low-resolution sources; skip source shared by `nodes` frames) inside the kernel — against
    float64 on torch's own interpolation, and against gcpx_conv_stage + gcpx_wgrad_conv3x3_split on the same data (the path it
    replaces: same operand values, same scales, same sums).  Every tile shape (W = 64 / 32 / 16 / 8)."""
    if tiles == "2x32":          # 64-pixel tiles as 2 rows x 32 columns (the form before round 6; images of 32+ columns)
        monkeypatch.setenv("GCPX_WS_TW32", "1")
    rt, pk, lib, dev = env
    torch.manual_seed(Hin + c_prev + c_skip)
    Cout, S = 16, 2 * Hin
    x = torch.randn(Fr, c_prev, Hin, Hin)
    sc, sh = torch.rand(c_prev) + 0.5, torch.randn(c_prev) * 0.2
    ref = [F.leaky_relu(x * sc[None, :, None, None] + sh[None, :, None, None], 0.2)]
    srcs = [(x.permute(0, 2, 3, 1).contiguous().to(dev), c_prev, 1, sc.to(dev), sh.to(dev), rt.ACT_LRELU)]
    if c_skip:
        sk = torch.randn(Fr // nodes, c_skip, Hin, Hin)
        ref.append(sk.repeat_interleave(nodes, 0))
        srcs.append((sk.permute(0, 2, 3, 1).contiguous().to(dev), c_skip, nodes, None, None, rt.ACT_NONE))
    cin = c_prev + c_skip
    U = F.interpolate(torch.cat(ref, 1).double(), scale_factor=2, mode="bilinear", align_corners=False)      # [Fr][cin][S][S]
    dy = torch.randn(Fr, S, S, Cout, device=dev) * (1e-6 if case == "tiny" else 1.0)
    up = F.pad(U.permute(0, 2, 3, 1), (0, 0, 1, 1, 1, 1))
    want = torch.stack([torch.einsum("fhwn,fhwc->nc", dy.cpu().double(), up[:, ky:ky + S, kx:kx + S, :]) for ky in range(3) for kx in range(3)], 1)
    want = want.reshape(Cout, 9 * cin)
    grid = max(1, min(96 // (cin // 32), Fr * max(1, S * S // 64)))
    staged = torch.full((Fr, S, S, cin), float("nan"), device=dev)
    a = _conv_args(rt, srcs, F=Fr, Hin=Hin, Win=Hin, Hout=S, Wout=S, Cout=cin, out_pitch=cin, upsample=1, out=staged)
    rt.check(lib.gcpx_conv_stage(C.byref(a), _stream()), "conv_stage")
    p_ref = torch.full((grid, Cout, 9 * cin), float("nan"), device=dev)
    rt.check(lib.gcpx_wgrad_conv3x3_split(dy.data_ptr(), Cout, staged.data_ptr(), Fr, S, S, cin, Cout, p_ref.data_ptr(), grid, _stream()), "staged")
    p_up = torch.full((grid, Cout, 9 * cin), float("nan"), device=dev)
    a.out = None
    rt.check(lib.gcpx_wgrad_conv3x3_split_up(dy.data_ptr(), Cout, C.byref(a), Cout, p_up.data_ptr(), grid, _stream()), "fused")
    torch.cuda.synchronize()
    assert torch.isfinite(p_up).all()
    got, ref_ = p_up.double().sum(0).cpu(), p_ref.double().sum(0).cpu()
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) <= 2.0 * float((ref_ - want).abs().max()) + 4e-7 * scale
    assert_close(got.float(), ref_.float(), atol=2e-6 * scale, rtol=0, name="fused vs staged")


@pytest.mark.parametrize("tiles", ["4x16", "2x32"])
@pytest.mark.parametrize("Cout,Cin,S", [(100, 16, 64), (16, 32, 32), (32, 64, 16)])
def test_wgrad_conv3x3_split_src_applies_affine_and_frame_map(env, Cout, Cin, S, tiles, monkeypatch):
    """gcpx_wgrad_conv3x3_split_src: the operand is LeakyReLU(scale x + shift) of a raw tensor read through a frame map (the output
    head's weight gradient over the matched rows) — against gcpx_conv_stage (gathered, activated copy) + gcpx_wgrad_conv3x3_split, the
    path it replaces (same operand values: identical sums), and against float64."""
    if tiles == "2x32":          # 64-pixel tiles as 2 rows x 32 columns (the form before round 6; images of 32+ columns)
        monkeypatch.setenv("GCPX_WS_TW32", "1")
    rt, pk, lib, dev = env
    torch.manual_seed(Cout + S)
    Fx, R = 9, 6
    N16 = (Cout + 15) // 16 * 16
    x = torch.randn(Fx, S, S, Cin, device=dev)
    sc, sh = (torch.rand(Cin) + 0.5).to(dev), (torch.randn(Cin) * 0.3).to(dev)
    fmap = torch.tensor([4, 0, 8, 8, 2, 5], dtype=torch.int32, device=dev)
    dy = torch.randn(R, S, S, N16, device=dev) * 1e-3
    dy[..., Cout:] = 0.0
    staged = torch.full((R, S, S, Cin), float("nan"), device=dev)
    a = _conv_args(rt, [(x, Cin, 1, sc, sh, rt.ACT_LRELU)], F=R, Hin=S, Win=S, Hout=S, Wout=S, Cout=Cin, out_pitch=Cin, upsample=0, out=staged)
    a.src_row_map = fmap.data_ptr()
    rt.check(lib.gcpx_conv_stage(C.byref(a), _stream()), "conv_stage")
    ych = Cin // 32 if (Cin % 32 == 0 and N16 != 112) else Cin // 16
    grid = max(1, min(96 // ych, R * max(1, S * S // 64)))
    p_ref = torch.full((grid, N16, 9 * Cin), float("nan"), device=dev)
    rt.check(lib.gcpx_wgrad_conv3x3_split(dy.data_ptr(), N16, staged.data_ptr(), R, S, S, Cin, Cout, p_ref.data_ptr(), grid, _stream()), "staged")
    p_src = torch.full((grid, N16, 9 * Cin), float("nan"), device=dev)
    bpart = torch.full((grid, N16), float("nan"), device=dev) if N16 == 112 else None       # the head form also sums dy's columns
    rt.check(lib.gcpx_wgrad_conv3x3_split_src(dy.data_ptr(), N16, x.data_ptr(), fmap.data_ptr(), sc.data_ptr(), sh.data_ptr(), R, S, S, Cin, Cout,
                                              p_src.data_ptr(), (bpart.data_ptr() if bpart is not None else None), grid, _stream()), "src")
    torch.cuda.synchronize()
    if bpart is not None:
        wantb = dy.double().sum((0, 1, 2))
        assert_close(bpart.double().sum(0).float(), wantb.float(), atol=3e-6 * float(wantb.abs().max()), rtol=0, name="bias column sums")
    u = F.leaky_relu(x.double() * sc.double() + sh.double(), 0.2)[fmap.long()]
    up = F.pad(u, (0, 0, 1, 1, 1, 1))
    want = torch.stack([torch.einsum("fhwn,fhwc->nc", dy.double(), up[:, ky:ky + S, kx:kx + S, :]) for ky in range(3) for kx in range(3)], 1)
    want = want.reshape(N16, 9 * Cin)
    assert torch.isfinite(p_src).all()
    got, ref_ = p_src.double().sum(0), p_ref.double().sum(0)
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) <= 2.0 * float((ref_ - want).abs().max()) + 4e-7 * scale
    assert_close(got.float(), ref_.float(), atol=2e-6 * scale, rtol=0, name="src vs staged")


def test_split_pack_group_equals_single_launches(env):
    """gcpx_split_pack_group (all split-f16 weight tensors re-split in one launch, one workgroup per tensor) writes bit for bit what one
    gcpx_split_pack launch per tensor writes: tensors of different sizes and magnitudes, one of them all zero."""
    rt, pk, lib, dev = env
    torch.manual_seed(4)
    shapes = [((100, 16, 3, 3), 1.0), ((16, 32, 3, 3), 1e-3), ((32, 64, 3, 3), 30.0), ((16, 32, 3, 3), 0.0)]
    off, parts, idxs = 8, [torch.randn(8) * 100], []
    for shp, amp in shapes:
        w = torch.randn(*shp) * amp
        idxs.append(pk.conv3x3_split_index(shp, off, None).to(dev))
        parts.append(w.reshape(-1))
        off += w.numel()
    theta = torch.cat(parts).to(dev)
    outs_a = [torch.full((2 * i.numel(),), -1, dtype=torch.int16, device=dev) for i in idxs]
    outs_b = [torch.full((2 * i.numel(),), -2, dtype=torch.int16, device=dev) for i in idxs]
    ea, eb = torch.full((len(idxs),), 99, dtype=torch.int32, device=dev), torch.full((len(idxs),), 98, dtype=torch.int32, device=dev)
    for k, i in enumerate(idxs):
        rt.check(lib.gcpx_split_pack(theta.data_ptr(), i.data_ptr(), i.numel(), outs_a[k].data_ptr(), ea.data_ptr() + 4 * k, _stream()), "split_pack")
    descs = []
    for k, i in enumerate(idxs):
        d = rt.SplitPackDesc()
        d.src, d.idx, d.out, d.log2_out, d.n = theta.data_ptr(), i.data_ptr(), outs_b[k].data_ptr(), eb.data_ptr() + 4 * k, i.numel()
        descs.append(d)
    arr = (rt.SplitPackDesc * len(descs))(*descs)
    tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
    rt.check(lib.gcpx_split_pack_group(tab.data_ptr(), len(descs), _stream()), "split_pack_group")
    torch.cuda.synchronize()
    assert torch.equal(ea, eb)
    for a_, b_ in zip(outs_a, outs_b):
        assert torch.equal(a_, b_)
    # ... and so does the two-launch form with 32 workgroups per tensor (the one the trainer runs behind every optimizer step); twice,
    # the second time over different parameters: its scratch must not carry the first call's maxima over
    for rep in range(2):
        if rep:
            theta.mul_(0.25)
            for k, i in enumerate(idxs):
                rt.check(lib.gcpx_split_pack(theta.data_ptr(), i.data_ptr(), i.numel(), outs_a[k].data_ptr(), ea.data_ptr() + 4 * k, _stream()), "split_pack")
        for o in outs_b:
            o.fill_(-3)
        eb.fill_(97)
        scratch = torch.full((len(descs),), 0x7f7fffff, dtype=torch.int32, device=dev)
        rt.check(lib.gcpx_split_pack_group2(tab.data_ptr(), len(descs), scratch.data_ptr(), _stream()), "split_pack_group2")
        torch.cuda.synchronize()
        assert torch.equal(ea, eb), rep
        for a_, b_ in zip(outs_a, outs_b):
            assert torch.equal(a_, b_), rep


# ---- the split-f16 bound, stated as it is (norm-wise per scaled unit) and probed where it is weakest -------------------------------
#
# A value x of a unit that shares one power-of-two scale 2^E (2^14 <= max|x| 2^E < 2^15: an item of the head, a staged block of frames of
# an encoder layer, a row of the GEMM from the stage on that holds the row's largest magnitude, a whole weight tensor) is kept as two f16
# pieces whose sum differs from x by at most
#         max( 2^-22 |x| ,  2^-39 max|x|_unit )          (the second term: the f16 subnormal quantum 2^-24 / 2^-25 under the scale),
# and the product x w adds the dropped x2 w2 <= 2^-22 |x w|.  Element-wise the result of a split kernel is therefore off by at most
#         sum_k [ 3 * 2^-22 |x_k w_k| + 2^-39 ( Xmax |w_k| + |x_k| Wmax ) ]   + the f32 accumulation's own roundings,
# which is one f32 rounding of the result when the terms near the unit's maximum carry it (everything the decoder sees behind a
# BatchNorm), and NOT when a large channel meets zero weights beside a small channel that carries the output: values 2^-k of the unit's
# maximum keep about 39 - k bits.  The cases below build exactly that and report the element-wise relative error next to the exact kernel's.
_SPLIT_C1, _SPLIT_C2 = 4 * 2.0 ** -22, 2.0 ** -38


def _split_bound(absx_dot_absw, xmax, sum_absw, wmax, sum_absx):
    return _SPLIT_C1 * absx_dot_absw + _SPLIT_C2 * (xmax * sum_absw + wmax * sum_absx)


def _report_split_bound(name, err, dot, bound):
    """element-wise error relative to sum_k |x_k w_k| of the SAME element (not to the largest result: an output that cancels is not the
    point, one whose terms are all small beside a large neighbour is)"""
    rel = {k: float((v / dot.clamp_min(1e-300)).max()) for k, v in err.items()}
    frac = float((err["split"] / bound).max())
    print(f"SPLIT-BOUND {name}: max over elements of |error| / sum_k |x_k w_k|: split {rel['split']:.3e}, exact f32 {rel['f32']:.3e}; "
          f"split error / stated bound (max over elements) {frac:.3f}")
    assert frac <= 1.0, (name, frac)
    return rel


@pytest.mark.parametrize("form", ["split", "planes"])
@pytest.mark.parametrize("ratio_log2", [20, 31])
def test_split_bound_gemm_large_column_with_zero_weights(env, ratio_log2, form):
    """Row GEMM: the first 64 columns of every row hold values 2^ratio_log2 times larger than the rest and meet ZERO weights; the small
    columns carry the whole output.  The row's scale is pinned by the large values (split kernel: from the first stage on; two-launch
    planes form: one scale per row), so the small ones keep ~39 - ratio_log2 bits: the result obeys the stated norm-wise bound and is
    element-wise far from f32-equivalent."""
    rt, pk, lib, dev = env
    torch.manual_seed(ratio_log2)
    M, N, K = 4608, 256, 512                                     # (more than 4096 rows: smaller problems stay on the exact kernel's blocks)
    small = 2.0 ** -10
    x = torch.randn(M, K) * small
    x[:, :64] = small * 2.0 ** ratio_log2 * (1 + torch.rand(M, 64))
    w, b = torch.randn(N, K) / K ** 0.5, torch.zeros(N)
    w[:, :64] = 0.0
    ref = F.linear(x.double(), w.double())
    xd, wp, bd = x.to(dev), pk.pack_gemm(w).to(dev), b.to(dev)
    ws, e = pk.pack_gemm_split(w)
    ws = ws.to(dev)
    err = {}
    for name in ("f32", "split"):
        out = torch.full((M, N), float("nan"), device=dev)
        a = rt.GemmArgs()
        a.src[0] = _rowsrc(rt, xd, 0, K, K)
        a.nsrc, a.M, a.N, a.K, a.rpb = 1, M, N, K, M
        a.wpk, a.bias, a.out, a.ob, a.orow = wp.data_ptr(), bd.data_ptr(), out.data_ptr(), 0, N
        if name == "split":
            a.wpk_split, a.w_split_log2 = ws.data_ptr(), e
            if form == "planes":
                keep = _planes_workspace(rt, lib, a, dev)
        rt.check(lib.gcpx_gemm(C.byref(a), _stream()), name)
        torch.cuda.synchronize()
        err[name] = (out.cpu().double() - ref).abs()
    ax, aw = x.double().abs(), w.double().abs()
    bound = _split_bound(ax @ aw.T, ax.amax(1, keepdim=True), aw.sum(1)[None, :], float(aw.max()), ax.sum(1, keepdim=True)) \
        + 2.0 ** -22 * (ax @ aw.T)                               # the f32 accumulation itself (K = 512 terms)
    rel = _report_split_bound(f"gemm ({form}) 4608x256x512, large/small = 2^{ratio_log2}", err, ax @ aw.T, bound)
    assert rel["f32"] < 1e-5                                    # the exact kernel does not care
    if ratio_log2 >= 31:
        assert rel["split"] > 10 * rel["f32"]                   # ... the split kernel does: this is what "norm-wise" means


@pytest.mark.parametrize("ratio_log2", [20, 31])
def test_split_bound_head_large_channel_with_zero_weights(env, ratio_log2):
    """Output head: channel 0 of the last decoder block is 2^ratio_log2 times larger than the other 15 and has zero weights; every
    4 x 16-pixel item's scale is pinned by it."""
    rt, pk, lib, dev = env
    torch.manual_seed(40 + ratio_log2)
    S, Fr = 64, 2
    small = 2.0 ** -10
    x = torch.randn(Fr, 16, S, S) * small
    x[:, 0] = small * 2.0 ** ratio_log2 * (1 + torch.rand(Fr, S, S))
    sc, sh = torch.ones(16), torch.zeros(16)
    xin = F.leaky_relu(x, 0.2)
    w, b = torch.randn(100, 16, 3, 3) / 12.0, torch.zeros(100)
    w[:, 0] = 0.0
    ref = F.conv2d(xin.double(), w.double(), padding=1)
    perm = pk.dlm_channel_perm(10)
    permt = torch.tensor(perm)
    wp = pk.pack_dlm_head(w, perm).to(dev)
    ws, e = pk.pack_conv3x3_split(w, perm)
    ws = ws.to(dev)
    bk = torch.zeros(len(perm))
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    slots = torch.nonzero(permt >= 0)[:, 0]
    inv = torch.empty(100, dtype=torch.long)
    inv[permt[slots]] = slots
    err = {}
    for name in ("f32", "split"):
        raw = torch.full((Fr, S, S, len(perm)), float("nan"), device=dev)
        img = torch.full((Fr, 3, S, S), float("nan"), device=dev)
        a = _conv_args(rt, [(xd, 16, 1, sc.to(dev), sh.to(dev), rt.ACT_LRELU)], F=Fr, Hin=S, Win=S, Hout=S, Wout=S, Cout=100,
                       out_pitch=len(perm), upsample=0, head_mode=rt.HEAD_DLM_BOTH, wpk=wp, bias=bk.to(dev), out=raw, images=img)
        if name == "split":
            a.wpk_split, a.w_split_log2 = ws.data_ptr(), e
        rt.check(lib.gcpx_conv3x3(C.byref(a), _stream()), name)
        torch.cuda.synchronize()
        err[name] = (raw.cpu().index_select(-1, inv).permute(0, 3, 1, 2).double() - ref).abs()
    ax, aw = xin.double().abs(), w.double().abs()
    dot = F.conv2d(ax, aw, padding=1)
    sum_absx = F.conv2d(ax, torch.ones(1, 16, 3, 3, dtype=torch.float64), padding=1)            # [Fr, 1, S, S]
    bound = _split_bound(dot, float(ax.max()), aw.sum((1, 2, 3))[None, :, None, None], float(aw.max()), sum_absx) + 2.0 ** -22 * dot
    rel = _report_split_bound(f"head 16->100 @64x64, large/small = 2^{ratio_log2}", err, dot, bound)
    assert rel["f32"] < 1e-5
    if ratio_log2 >= 31:
        assert rel["split"] > 10 * rel["f32"]


@pytest.mark.parametrize("ratio_log2", [20, 31])
def test_split_bound_encoder_large_channel_with_zero_weights(env, ratio_log2):
    """First split layer of the encoder (16 -> 32 channels @32x32 -> 16x16, un-normalised inputs): same construction; one scale per
    staged block of frames."""
    rt, pk, lib, dev = env
    torch.manual_seed(80 + ratio_log2)
    Hin, cin, cout, Fr = 32, 16, 32, 3
    small = 2.0 ** -10
    x = torch.randn(Fr, cin, Hin, Hin) * small
    x[:, 0] = small * 2.0 ** ratio_log2 * (1 + torch.rand(Fr, Hin, Hin))
    sc, sh = torch.ones(cin), torch.zeros(cin)
    xin = F.leaky_relu(x, 0.2)
    w, b = torch.randn(cout, cin, 4, 4) / (16 * cin) ** 0.5, torch.zeros(cout)
    w[:, 0] = 0.0
    ref = F.conv2d(xin.double(), w.double(), stride=2, padding=1)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wp, bd = pk.pack_conv4x4(w).to(dev), b.to(dev)
    ws, e = pk.pack_conv4x4_split(w)
    ws = ws.to(dev)
    err = {}
    for name in ("f32", "split"):
        out = torch.full((Fr, Hin // 2, Hin // 2, cout), float("nan"), device=dev)
        a = _conv_args(rt, [(xd, cin, 1, sc.to(dev), sh.to(dev), rt.ACT_LRELU)], F=Fr, Hin=Hin, Win=Hin, Hout=Hin // 2, Wout=Hin // 2,
                       Cout=cout, out_pitch=cout, wpk=wp, bias=bd, out=out)
        if name == "split":
            a.wpk_split, a.w_split_log2 = ws.data_ptr(), e
        rt.check(lib.gcpx_conv4x4s2(C.byref(a), _stream()), name)
        torch.cuda.synchronize()
        err[name] = (out.cpu().permute(0, 3, 1, 2).double() - ref).abs()
    ax, aw = xin.double().abs(), w.double().abs()
    dot = F.conv2d(ax, aw, stride=2, padding=1)
    sum_absx = F.conv2d(ax, torch.ones(1, cin, 4, 4, dtype=torch.float64), stride=2, padding=1)
    bound = _split_bound(dot, float(ax.max()), aw.sum((1, 2, 3))[None, :, None, None], float(aw.max()), sum_absx) + 2.0 ** -22 * dot
    rel = _report_split_bound(f"encoder 16->32 @32x32, large/small = 2^{ratio_log2}", err, dot, bound)
    assert rel["f32"] < 1e-5
    if ratio_log2 >= 31:
        assert rel["split"] > 10 * rel["f32"]


# ---- the row GEMM as conversion pass + LDS-DMA fed GEMM (csrc/gemm_planes.hip) ---------------------------------------------------------
def _planes_workspace(rt, lib, a, dev):
    """hands gcpx_gemm the activation-planes workspace: with it (and wpk_split) problems from 512 rows on take the two-launch form"""
    nbytes, nexp = C.c_int64(), C.c_int64()
    rt.check(lib.gcpx_gemm_planes_workspace(a.M, a.K, a.nbatch, C.byref(nbytes), C.byref(nexp)), "workspace")
    planes = torch.full((nbytes.value // 2,), float("nan"), dtype=torch.float16, device=dev)
    exps = torch.full((nexp.value,), 77, dtype=torch.int32, device=dev)
    a.x_planes, a.x_exp, a.x_planes_bytes = planes.data_ptr(), exps.data_ptr(), nbytes.value
    return planes, exps


@pytest.mark.parametrize("case", ["unit", "tiny", "row_scales", "k_growth", "outlier", "zero_rows"])
@pytest.mark.parametrize("M,N,K", [(1024, 2048, 1024), (8200, 1024, 192), (33000, 2048, 256), (2100, 2048, 128)])
def test_gemm_planes_error_vs_float64(env, case, M, N, K):
    """The two-launch form of the split-f16 row GEMM (conversion pass with ONE power-of-two scale per row over all of K, then the GEMM fed by
    LDS-DMA) against float64 next to the exact f32 MFMA kernel, same cases and bounds as test_gemm_split_error_vs_float64.  The four
    shapes take the four tile configurations (64 x 128 / 128 x 256 / 256 x 256 / 128 x 128), three of them with a masked last row block
    (shapes the dispatcher keeps on the exact kernel's split-K blocks — few rows x few columns — never reach this form)."""
    rt, pk, lib, dev = env
    torch.manual_seed(M + K)
    x = torch.randn(M, K)
    if case == "tiny":
        x *= 1e-6
    elif case == "row_scales":
        x *= torch.logspace(-6, 3, M)[:, None]
    elif case == "k_growth":
        x *= torch.logspace(-2, 2, K)[None, :]
    elif case == "outlier":
        x[3, K // 2] = 3e4
    elif case == "zero_rows":
        x[::3] = 0.0
    w, b = torch.randn(N, K) / K ** 0.5, torch.randn(N)
    ref = F.linear(x.double(), w.double(), b.double())
    xd, wp, bd = x.to(dev), pk.pack_gemm(w).to(dev), b.to(dev)
    ws, e = pk.pack_gemm_split(w)
    ws = ws.to(dev)
    err = {}
    for name in ("f32", "planes"):
        out = torch.full((M, N), float("nan"), device=dev)
        a = rt.GemmArgs()
        a.src[0] = _rowsrc(rt, xd, 0, K, K)
        a.nsrc, a.M, a.N, a.K, a.rpb = 1, M, N, K, M
        a.wpk, a.bias, a.out, a.ob, a.orow = wp.data_ptr(), bd.data_ptr(), out.data_ptr(), 0, N
        if name == "planes":
            a.wpk_split, a.w_split_log2 = ws.data_ptr(), e
            keep = _planes_workspace(rt, lib, a, dev)
        rt.check(lib.gcpx_gemm(C.byref(a), _stream()), name)
        torch.cuda.synchronize()
        assert torch.isfinite(out).all()
        if name == "planes":
            assert not bool((keep[1][:M] == 77).any())             # the conversion pass ran (every row got its exponent)
        err[name] = (out.cpu().double() - ref).abs()
    rs = ref.abs().amax(1) + 1e-30
    rms = {k: v.pow(2).mean(1).sqrt() for k, v in err.items()}
    # k_growth: one scale per row over all of K (the early, small columns sit 2^-13 below the row's maximum: still 22 bits) — same bounds
    assert bool((rms["planes"] <= 1.5 * rms["f32"] + 1e-7 * rs).all()), float((rms["planes"] / (rms["f32"] + 1e-7 * rs)).max())
    assert bool((err["planes"].amax(1) <= 2.0 * err["f32"].amax(1) + 4e-7 * rs).all())


def test_gemm_planes_lstm_sources_and_batches(env):
    """the two-launch form with what the tree levels use: concatenated sources with a row gather, the LSTM-cell epilogue (h, c, dense h
    copy), the conv1d-over-time form (shifted, masked sources with affine + LReLU on load) and batched problems with their own weight
    exponents"""
    rt, pk, lib, dev = env
    torch.manual_seed(19)
    H = 128
    pool = torch.randn(500, H)
    cell = torch.nn.LSTMCell(H, H)
    w, b = pk.lstm_gate_interleave(cell.weight_ih.detach(), cell.weight_hh.detach(), cell.bias_ih.detach(), cell.bias_hh.detach())
    wp, bd = pk.pack_gemm(w).to(dev), b.to(dev)
    ws, e = pk.pack_gemm_split(w)
    ws = ws.to(dev)
    pd = pool.to(dev)
    for Mb in (4200, 16500):
        ridx_b = torch.randint(0, 500, (Mb,), dtype=torch.int32)
        xb_, hb_, cb_ = pool[ridx_b.long()], torch.randn(Mb, H), torch.randn(Mb, H)
        with torch.no_grad():
            h1b, c1b = cell(xb_, (hb_, cb_))
        rdb, hdb, cdb = ridx_b.to(dev), hb_.to(dev), cb_.to(dev)
        hob, cob, hcb = (torch.full((Mb, H), float("nan"), device=dev) for _ in range(3))
        a = rt.GemmArgs()
        a.src[0] = _rowsrc(rt, pd, 0, H, H, rowidx=rdb)
        a.src[1] = _rowsrc(rt, hdb, 0, H, H)
        a.nsrc, a.M, a.N, a.K, a.rpb = 2, Mb, 4 * H, 2 * H, Mb
        a.wpk, a.bias, a.epi = wp.data_ptr(), bd.data_ptr(), rt.EPI_LSTM
        a.wpk_split, a.w_split_log2 = ws.data_ptr(), e
        a.c_prev, a.c_prev_stride, a.h_out, a.c_out, a.hb, a.hrow, a.h_copy = cdb.data_ptr(), H, hob.data_ptr(), cob.data_ptr(), 0, H, hcb.data_ptr()
        # gates_out (the training forward keeps the activated gates for the backward pass): written by the planes form's epilogue ...
        gob = torch.full((Mb, H, 4), float("nan"), device=dev)
        a.gates_out = gob.data_ptr()
        with torch.no_grad():
            pre = (F.linear(xb_, cell.weight_ih, cell.bias_ih) + F.linear(hb_, cell.weight_hh, cell.bias_hh)).view(Mb, 4, H)
            gates_ref = torch.stack([torch.sigmoid(pre[:, 0]), torch.sigmoid(pre[:, 1]), torch.tanh(pre[:, 2]), torch.sigmoid(pre[:, 3])], 2)
        keep = _planes_workspace(rt, lib, a, dev)
        rt.check(lib.gcpx_gemm(C.byref(a), _stream()), "gemm planes lstm")
        torch.cuda.synchronize()
        assert not bool((keep[1][:Mb] == 77).any())
        assert_close(hob, h1b, atol=1e-5, name=f"h, {Mb} rows")
        assert_close(cob, c1b, atol=1e-5, name=f"c, {Mb} rows")
        assert_close(hcb, h1b, atol=1e-5, name=f"h_copy, {Mb} rows")
        assert_close(gob, gates_ref, atol=1e-5, name=f"gates, {Mb} rows")
        # ... and by the one-launch split form's (no workspace)
        a.x_planes, a.x_exp, a.x_planes_bytes = None, None, 0
        gob.fill_(float("nan")); hob.fill_(float("nan"))
        rt.check(lib.gcpx_gemm(C.byref(a), _stream()), "gemm split lstm")
        torch.cuda.synchronize()
        assert_close(hob, h1b, atol=1e-5, name=f"h (split), {Mb} rows")
        assert_close(gob, gates_ref, atol=1e-5, name=f"gates (split), {Mb} rows")
    # conv1d-over-time form
    B, T, Cc, N = 16, 40, 64, 128
    xs = torch.randn(B, T, Cc)
    sc, sh = torch.rand(Cc) + 0.5, torch.randn(Cc) * 0.1
    wc, bc = torch.randn(N, Cc, 3) / (3 * Cc) ** 0.5, torch.randn(N)
    want = F.leaky_relu(F.conv1d(F.leaky_relu(xs * sc + sh, 0.2).transpose(1, 2), wc, bc, padding=1).transpose(1, 2), 0.2)
    w2 = wc.permute(0, 2, 1).reshape(N, 3 * Cc)
    xd, scd, shd, bcd = xs.to(dev), sc.to(dev), sh.to(dev), bc.to(dev)
    wp2 = pk.pack_gemm(w2).to(dev)
    ws2, e2 = pk.pack_gemm_split(w2)
    ws2 = ws2.to(dev)
    out = torch.full((B * T, N), float("nan"), device=dev)
    a = rt.GemmArgs()
    for i, d in enumerate((-1, 0, 1)):
        a.src[i] = _rowsrc(rt, xd, T * Cc, Cc, Cc, shift=d, scale=scd, shiftv=shd, act=rt.ACT_LRELU, cmod=Cc)
    a.nsrc, a.M, a.N, a.K, a.rpb = 3, B * T, N, 3 * Cc, T
    a.wpk, a.bias, a.out, a.ob, a.orow, a.epi = wp2.data_ptr(), bcd.data_ptr(), out.data_ptr(), T * N, N, rt.EPI_LRELU
    a.wpk_split, a.w_split_log2 = ws2.data_ptr(), e2
    keep = _planes_workspace(rt, lib, a, dev)
    rt.check(lib.gcpx_gemm(C.byref(a), _stream()), "gemm planes conv1d")
    torch.cuda.synchronize()
    assert not bool((keep[1][:B * T] == 77).any())
    assert_close(out.view(B, T, N), want, atol=2e-5, rtol=1e-5, name="conv1d-gemm planes")
    # batched: 3 problems, weights of different magnitude (their own exponents), rows not a multiple of the tile
    nb, M2, N2, K2 = 3, 4200, 128, 256
    xb = torch.randn(nb, M2, K2)
    wb = torch.randn(nb, N2, K2) / K2 ** 0.5 * torch.tensor([1.0, 1e-3, 50.0])[:, None, None]
    bb = torch.randn(nb, N2)
    wantb = torch.einsum("bmk,bnk->bmn", xb, wb) + bb[:, None, :]
    packs = [pk.pack_gemm_split(wb[i]) for i in range(nb)]
    wsb = torch.stack([p_[0] for p_ in packs]).contiguous().to(dev)
    eb = torch.tensor([p_[1] for p_ in packs], dtype=torch.int32, device=dev)
    wpb = torch.stack([pk.pack_gemm(wb[i]) for i in range(nb)]).contiguous().to(dev)
    xbd, bbd = xb.to(dev), bb.to(dev)
    outb = torch.full((nb, M2, N2), float("nan"), device=dev)
    a = rt.GemmArgs()
    a.src[0] = _rowsrc(rt, xbd, 0, K2, K2)
    a.nsrc, a.M, a.N, a.K, a.rpb = 1, M2, N2, K2, M2
    a.wpk, a.bias, a.out, a.ob, a.orow = wpb.data_ptr(), bbd.data_ptr(), outb.data_ptr(), 0, N2
    a.nbatch, a.z_src_off, a.z_w_off, a.z_bias_off, a.z_out_off = nb, M2 * K2, N2 * K2, N2, M2 * N2
    a.wpk_split, a.w_split_log2_dev = wsb.data_ptr(), eb.data_ptr()
    keep = _planes_workspace(rt, lib, a, dev)
    rt.check(lib.gcpx_gemm(C.byref(a), _stream()), "gemm planes batched")
    torch.cuda.synchronize()
    wantb = torch.einsum("bmk,bnk->bmn", xb.double(), wb.double()) + bb[:, None, :].double()
    for i, wscale in enumerate((1.0, 1e-3, 50.0)):                # an f32 rounding of the terms' size, per problem
        assert_close(outb[i].double().cpu(), wantb[i], atol=3e-5 * max(wscale, 1.0), rtol=2e-5, name=f"batched planes, problem {i}")


@pytest.mark.parametrize("B,nodes,H,Ca,Cs", [(3, 5, 16, 16, 16), (2, 7, 8, 64, 64), (16, 3, 32, 16, 16)])
def test_act_skip_bwd_equals_the_two_passes(env, B, nodes, H, Ca, Cs):
    """gcpx_act_skip_bwd (one pass over the upsampling block's input gradient: previous-block half through the transposed bilinear x2,
    LeakyReLU' and the BatchNorm-backward sums; skip half summed over a sequence's frames) against the two gcpx_act_bwd launches it
    replaces in the decoder's backward, and the first half against torch's own bilinear backward."""
    rt, pk, lib, dev = env
    torch.manual_seed(B * H)
    Fr, ldc = B * nodes, Ca + Cs
    dU = torch.randn(Fr, 2 * H, 2 * H, ldc, device=dev)
    r = torch.randn(Fr, H, H, Ca, device=dev)
    sc, sh = torch.rand(Ca, device=dev) + 0.5, torch.randn(Ca, device=dev) * 0.2
    mean, rstd = torch.randn(Ca, device=dev) * 0.1, torch.rand(Ca, device=dev) + 0.5
    nb = lib.gcpx_act_bwd_blocks()

    def act_args(dy, st):
        a = rt.ActBwdArgs()
        a.da, a.r, a.scale, a.shift, a.mean, a.rstd = dU.data_ptr(), r.data_ptr(), sc.data_ptr(), sh.data_ptr(), mean.data_ptr(), rstd.data_ptr()
        a.dy, a.stats_partial, a.ldc, a.c_off, a.up, a.fsum, a.act = dy.data_ptr(), st.data_ptr(), ldc, 0, 1, 1, rt.ACT_LRELU
        a.F, a.H, a.W, a.C = Fr, H, H, Ca
        return a
    dy0, st0 = torch.full((Fr, H, H, Ca), float("nan"), device=dev), torch.full((nb, 2, Ca), float("nan"), device=dev)
    a0 = act_args(dy0, st0)
    rt.check(lib.gcpx_act_bwd(C.byref(a0), _stream()), "act")
    ds0 = torch.full((B, H, H, Cs), float("nan"), device=dev)
    a1 = rt.ActBwdArgs()
    a1.da, a1.dy, a1.ldc, a1.c_off, a1.up, a1.fsum, a1.act = dU.data_ptr(), ds0.data_ptr(), ldc, Ca, 1, nodes, rt.ACT_NONE
    a1.F, a1.H, a1.W, a1.C = B, H, H, Cs
    rt.check(lib.gcpx_act_bwd(C.byref(a1), _stream()), "skip")
    dy1, st1 = torch.full((Fr, H, H, Ca), float("nan"), device=dev), torch.full((nb, 2, Ca), float("nan"), device=dev)
    ds1 = torch.full((B, H, H, Cs), float("nan"), device=dev)
    a2 = act_args(dy1, st1)
    rt.check(lib.gcpx_act_skip_bwd(C.byref(a2), ds1.data_ptr(), Ca, Cs, nodes, _stream()), "act+skip")
    torch.cuda.synchronize()
    assert torch.equal(dy1, dy0) and torch.equal(ds1, ds0)          # same taps in the same order, frames summed in order
    assert_close(st1.sum(0).cpu(), st0.sum(0).cpu(), atol=2e-3 * float(st0.sum(0).abs().max()) + 1e-3, name="BatchNorm-backward sums")
    # the previous-block half against autograd through F.interpolate
    x = torch.zeros(Fr, Ca, H, H, dtype=torch.float64, requires_grad=True)
    up = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
    up.backward(dU[..., :Ca].permute(0, 3, 1, 2).double().cpu())
    slope = torch.where((r * sc + sh) > 0, 1.0, 0.2).cpu().double()
    assert_close(dy1.cpu().double(), x.grad.permute(0, 2, 3, 1) * slope, atol=1e-5, name="transposed bilinear x LeakyReLU'")


@pytest.mark.parametrize("M,N,K", [(8192, 2048, 256), (1100, 2048, 1024), (4200, 1024, 512)])
def test_gemm_planes_same_bits_under_memory_load(env, M, N, K):
    """Race screen of the LDS-DMA pipeline (hand-counted waits, one barrier per stage): the same problem 24 times while a second stream
    streams 1 GB copies through the chip — an early fragment read or a stage restaged too soon shows up as a run whose bits differ.
    Every run must reproduce the first one exactly, and the first one the float64 result."""
    rt, pk, lib, dev = env
    torch.manual_seed(M)
    x = torch.randn(M, K)
    w, b = torch.randn(N, K) / K ** 0.5, torch.randn(N)
    ref = F.linear(x.double(), w.double(), b.double())
    xd, wp, bd = x.to(dev), pk.pack_gemm(w).to(dev), b.to(dev)
    ws, e = pk.pack_gemm_split(w)
    ws = ws.to(dev)
    a = rt.GemmArgs()
    a.src[0] = _rowsrc(rt, xd, 0, K, K)
    a.nsrc, a.M, a.N, a.K, a.rpb = 1, M, N, K, M
    out = torch.empty(M, N, device=dev)
    a.wpk, a.bias, a.out, a.ob, a.orow = wp.data_ptr(), bd.data_ptr(), out.data_ptr(), 0, N
    a.wpk_split, a.w_split_log2 = ws.data_ptr(), e
    keep = _planes_workspace(rt, lib, a, dev)
    hog_a, hog_b = torch.empty(1 << 28, dtype=torch.float32, device=dev), torch.empty(1 << 28, dtype=torch.float32, device=dev)
    side = torch.cuda.Stream()
    first = None
    for it in range(24):
        with torch.cuda.stream(side):
            if it % 3:
                hog_b.copy_(hog_a)                                 # uneven load: two runs of three beside a 1 GB copy, one alone
        out.fill_(float("nan"))
        rt.check(lib.gcpx_gemm(C.byref(a), _stream()), "planes")
        torch.cuda.synchronize()
        if first is None:
            first = out.clone()
            err = (first.double().cpu() - ref).abs().max()
            assert float(err) <= 2e-5 * float(ref.abs().max()), float(err)
        else:
            assert torch.equal(out, first), (it, int((out != first).sum()))


def test_randn_in_plan_generator(env):
    """gcpx_randn (Philox4x32-10 + Box-Muller; the latent noise of a forward that is not fed any): standard-normal moments and tails on
    4 M numbers, the same numbers for the same {seed, offset}, fresh numbers on every call (the offset advances on the device), other
    numbers for another seed, and sizes that are not a multiple of four."""
    rt, pk, lib, dev = env
    n = 1 << 22
    st = torch.tensor([1234, 0], dtype=torch.int64, device=dev)
    a, b, c = (torch.full((n,), float("nan"), device=dev) for _ in range(3))
    rt.check(lib.gcpx_randn(a.data_ptr(), n, st.data_ptr(), _stream()), "randn")
    rt.check(lib.gcpx_randn(b.data_ptr(), n, st.data_ptr(), _stream()), "randn")
    torch.cuda.synchronize()
    assert int(st[1]) == 2 * (n // 4) and int(st[0]) == 1234
    st2 = torch.tensor([1234, 0], dtype=torch.int64, device=dev)
    rt.check(lib.gcpx_randn(c.data_ptr(), n, st2.data_ptr(), _stream()), "randn")
    torch.cuda.synchronize()
    assert torch.equal(a, c) and not torch.equal(a, b)
    x = torch.cat([a, b]).double().cpu()
    assert torch.isfinite(x).all()
    assert abs(float(x.mean())) < 2e-3 and abs(float(x.var()) - 1.0) < 3e-3
    assert abs(float((x ** 3).mean())) < 1e-2 and abs(float((x ** 4).mean()) - 3.0) < 3e-2
    for thr, p in ((1.0, 0.158655), (2.0, 0.0227501), (3.0, 0.0013499)):
        frac = float((x > thr).double().mean())
        assert abs(frac - p) < 4 * (p / x.numel()) ** 0.5 + 1e-5, (thr, frac, p)
    assert abs(float((a[:-1].double() * a[1:].double()).mean().cpu())) < 3e-3            # neighbours uncorrelated
    st3 = torch.tensor([99, 0], dtype=torch.int64, device=dev)
    d = torch.full((1003 + 5,), 7.0, device=dev)
    rt.check(lib.gcpx_randn(d.data_ptr(), 1003, st3.data_ptr(), _stream()), "randn")
    torch.cuda.synchronize()
    assert bool((d[1003:] == 7.0).all()) and not bool((d[:1003] == 7.0).any()) and not torch.equal(d[:1003], a[:1003])
