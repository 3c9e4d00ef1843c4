"""Host-side weight packing: the fragment-order tensors must hold exactly the documented elements."""
import torch

from video_gcp_amd import packing as pk


def test_pack_gemm_layout():
    torch.manual_seed(0)
    w = torch.randn(48, 64)
    p = pk.pack_gemm(w)
    assert p.shape == (4, 3, 64, 4)
    for kg, nt, lane, s in [(0, 0, 0, 0), (3, 2, 63, 3), (1, 1, 17, 2), (2, 0, 40, 1)]:
        n, k = nt * 16 + lane % 16, kg * 16 + (lane // 16) * 4 + s
        assert p[kg, nt, lane, s] == w[n, k]


def test_pack_gemm_pads_rows():
    w = torch.randn(2, 32)
    p = pk.pack_gemm(w)
    assert p.shape == (2, 1, 64, 4)
    assert p[0, 0, 2, 0] == 0 and p[0, 0, 1, 0] == w[1, 0]


def test_pack_conv3x3_layout_and_padding():
    torch.manual_seed(1)
    w = torch.randn(20, 64, 3, 3)
    p = pk.pack_conv3x3(w, 32)
    assert p.shape == (2 * 9 * 2 + 1, 2, 64, 4)
    assert torch.all(p[-1] == 0)
    for chunk, tap, cgl, ct, lane, s in [(0, 0, 0, 0, 0, 0), (1, 8, 1, 1, 3, 3), (0, 4, 1, 0, 33, 2)]:
        co, ci = ct * 16 + lane % 16, chunk * 32 + cgl * 16 + (lane // 16) * 4 + s
        v = w[co, ci, tap // 3, tap % 3] if co < 20 else 0.0
        assert p[(chunk * 9 + tap) * 2 + cgl, ct, lane, s] == v
    assert p[0, 1, 4, 0] == 0      # co = 20: padding row


def test_dlm_perm_is_a_partial_permutation():
    perm = pk.dlm_channel_perm(10)
    assert len(perm) == 112
    real = [c for c in perm if c >= 0]
    assert sorted(real) == list(range(100))
    # slot 8k..8k+7 = logit_k, mean_{r,g,b}, coeff_{0,1,2}, log_scale_r; then log_scale_g, log_scale_b; 12 empty slots
    k = 3
    assert perm[8 * k:8 * k + 8] == [k, 10 + k, 40 + k, 70 + k, 30 + k, 60 + k, 90 + k, 20 + k]
    assert perm[100:] == [-1] * 12
    for kk in range(10):
        for c in range(3):
            assert perm[pk.dlm_log_scale_slot(c, kk)] == 10 + 30 * c + 10 + kk
    # green / blue log-scales of mixture 2 ct + h: channel tile 5, lane group 2 h + (ct >> 1), register pair ct & 1 (the lane that
    # evaluates the mixture in the head kernel's fused likelihood finds them after one row swap); mixtures 8, 9 = the 4-slot remainder
    assert [pk.dlm_log_scale_slot(1, kk) for kk in range(10)] == [80, 88, 82, 90, 84, 92, 86, 94, 96, 98]
    assert [pk.dlm_log_scale_slot(2, kk) for kk in range(10)] == [81, 89, 83, 91, 85, 93, 87, 95, 97, 99]


def test_pack_dlm_head_remainder_tile():
    torch.manual_seed(2)
    w = torch.randn(100, 16, 3, 3)
    perm = pk.dlm_channel_perm(10)
    p = pk.pack_dlm_head(w, perm)
    assert p.shape == (10, 7, 64, 4) and torch.all(p[-1] == 0)
    assert torch.equal(p[:, :6], pk.pack_conv3x3(w, 16, perm=perm)[:, :6])
    tap, kg, c, s_ = 5, 2, 3, 1
    assert p[tap, 6, kg * 4 + c, s_] == w[perm[96 + c], 4 * kg + s_, tap // 3, tap % 3]
    assert torch.all(p[:, 6, 16:] == 0)


def test_lstm_gate_interleave():
    H = 8
    w_ih, w_hh = torch.randn(4 * H, H), torch.randn(4 * H, H)
    b_ih, b_hh = torch.randn(4 * H), torch.randn(4 * H)
    w, b = pk.lstm_gate_interleave(w_ih, w_hh, b_ih, b_hh)
    u, g = 5, 2
    assert torch.equal(w[4 * u + g, :H], w_ih[g * H + u]) and torch.equal(w[4 * u + g, H:], w_hh[g * H + u])
    assert b[4 * u + g] == b_ih[g * H + u] + b_hh[g * H + u]


def test_pack_conv4x4_and_image():
    w = torch.randn(32, 16, 4, 4)
    p = pk.pack_conv4x4(w)
    assert p.shape == (16, 1, 2, 64, 4)
    tap, cg, ct, lane, s = 7, 0, 1, 37, 2
    assert p[tap, cg, ct, lane, s] == w[ct * 16 + lane % 16, (lane // 16) * 4 + s, tap // 4, tap % 4]
    wi = torch.randn(16, 3, 4, 4)
    q = pk.pack_conv4x4_image(wi)
    assert q.shape == (12, 1, 64)
    ci, ky, lane = 2, 3, 50
    assert q[ci * 4 + ky, 0, lane] == wi[lane % 16, ci, ky, lane // 16]


def test_split_f16_pieces_reconstruct_the_weight():
    """split_f16: w 2^e = w1 + w2 up to one f32 rounding, the largest magnitude just below the f16 maximum"""
    torch.manual_seed(3)
    for amp in (1.0, 1e-4, 50.0):
        w = torch.randn(64, 16, 3, 3) * amp
        w1, w2, e = pk.split_f16(w)
        assert w1.dtype == torch.float16 and w2.dtype == torch.float16
        ws = w.double() * 2.0 ** e
        assert 2.0 ** 14 <= float(ws.abs().max()) < 2.0 ** 15
        rel = ((ws - w1.double() - w2.double()).abs() / ws.abs().clamp_min(1e-300)).max()
        assert float(rel) <= 2.0 ** -22                       # two 11-bit pieces: 2^-22 worst case, ~2^-24 typical
        assert torch.isfinite(w1).all() and torch.isfinite(w2).all()
    w1, w2, e = pk.split_f16(torch.zeros(4, 16, 3, 3))
    assert e == 0 and not w1.any() and not w2.any()


def test_pack_conv3x3_split_layout():
    """[chunk][5 k-steps][CT][2 pieces][64 lanes][8]: lane (i, q) element el = W[16 ct + i][16 chunk + 8 (q & 1) + el][tap 2 s + (q >> 1)]"""
    torch.manual_seed(4)
    w = torch.randn(40, 32, 3, 3)
    p, e = pk.pack_conv3x3_split(w)
    assert p.shape == (2, 5, 3, 2, 64, 8) and p.dtype == torch.int16
    h = p.view(torch.float16).double()
    rec = (h[:, :, :, 0] + h[:, :, :, 1]) * 2.0 ** -e          # [chunk, s, ct, lane, el]
    for chunk, s, ct, lane, el in [(0, 0, 0, 0, 0), (1, 2, 1, 37, 5), (0, 4, 2, 17, 3), (1, 4, 0, 40, 1), (0, 3, 2, 63, 7)]:
        i, q = lane % 16, lane // 16
        co, ci, tap = 16 * ct + i, 16 * chunk + 8 * (q & 1) + el, 2 * s + (q >> 1)
        want = float(w[co, ci, tap // 3, tap % 3]) if co < 40 and tap < 9 else 0.0
        assert abs(float(rec[chunk, s, ct, lane, el]) - want) <= 2.0 ** -22 * abs(want)


def test_fold_up_weights_reproduces_upsample_conv():
    """packing.fold_up_weights: bilinear x2 (align_corners=False) + 3x3 conv (pad 1) == per output-row parity a 3x3 conv over the
    horizontally interpolated low-resolution rows (replicate rows, zero columns) with the folded weights, plus the -W0 / -W2
    corrections on the first / last output row (what conv3x3_up16_fold_kernel computes)."""
    import torch.nn.functional as F
    from video_gcp_amd import packing as pk
    torch.manual_seed(3)
    Fr, Cin, Cout, Hin, Win = 2, 32, 16, 4, 8
    x = torch.randn(Fr, Cin, Hin, Win, dtype=torch.float64)
    w = torch.randn(Cout, Cin, 3, 3)
    want = F.conv2d(F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False), w.double(), padding=1)
    fw = pk.fold_up_weights(w).double()                                     # [24, Cout, Cin]
    hx = F.interpolate(x, scale_factor=(1, 2), mode="bilinear", align_corners=False)        # horizontal half only
    hxp = F.pad(F.pad(hx, (1, 1, 0, 0)), (0, 0, 1, 1), mode="replicate")    # zero columns, replicate rows
    got = torch.zeros_like(want)
    for py in range(2):
        k = fw[py * 9:(py + 1) * 9].reshape(3, 3, Cout, Cin).permute(2, 3, 0, 1)             # [Cout, Cin, dyl, tx]
        got[:, :, py::2] = F.conv2d(hxp, k)
    ktop = fw[18:21].permute(1, 2, 0)[:, :, None, :]                        # [Cout, Cin, 1, tx]
    kbot = fw[21:24].permute(1, 2, 0)[:, :, None, :]
    got[:, :, 0:1] += F.conv2d(hxp[:, :, 0:1], ktop)
    got[:, :, -1:] += F.conv2d(hxp[:, :, -1:], kbot)
    assert float((got - want).abs().max()) < 1e-5 * float(want.abs().max())
    # fragment order: lane (i, q) element el = fw[t][i][8 q + el]; the index form addresses the same elements
    g = pk.conv3x3_fold_gather(pk.fold_up_weights(w))
    assert float(g[5, 16 * 2 + 3, 4]) == float(pk.fold_up_weights(w)[5, 3, 20])
    idx = pk.conv3x3_fold_index()
    assert torch.equal(pk.fold_up_weights(w).reshape(-1)[idx.long()].view(24, 64, 8), g)
