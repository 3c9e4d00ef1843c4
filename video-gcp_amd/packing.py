"""Weight packing into MFMA fragment order (done once at model build; pure index shuffling).

All hot kernels put output channels / GEMM columns on the MFMA i side, so the A operand of
v_mfma_f32_16x16x4_f32 is a weight tile:  lane l holds W[n = 16*nt + (l & 15)][k] for the 4 k-values
k = 16*kg + 4*(l >> 4) + s, s = 0..3 (one float4).  A packed tensor is therefore [steps][tiles][64 lanes][4]:
each (step, tile) is one coalesced 1 KiB wavefront load.  See csrc/common.h for the operand maps.
"""
import torch

_LANE = torch.arange(64)
_LI = _LANE % 16          # i (row inside the 16-wide tile)
_LQ = _LANE // 16         # kk


def _pad_rows(w, rows):
    if w.shape[0] == rows:
        return w
    pad = torch.zeros((rows - w.shape[0],) + tuple(w.shape[1:]), dtype=w.dtype, device=w.device)
    return torch.cat([w, pad], 0)


def pack_gemm(w):
    """w [N, K] (N % 16 == 0 after zero padding, K % 16 == 0) -> [K/16][N/16][64][4]."""
    N, K = w.shape
    assert K % 16 == 0, (N, K)
    NT = (N + 15) // 16
    w = _pad_rows(w, NT * 16)
    dev = w.device
    kg = torch.arange(K // 16, device=dev)[:, None, None, None]
    nt = torch.arange(NT, device=dev)[None, :, None, None]
    li = _LI.to(dev)[None, None, :, None]
    lq = _LQ.to(dev)[None, None, :, None]
    s = torch.arange(4, device=dev)[None, None, None, :]
    return w[nt * 16 + li, kg * 16 + lq * 4 + s].contiguous()


def unpack_gemm(wpk, N):
    """inverse of pack_gemm: [K/16][NT][64][4] -> w [N, K]"""
    KG, NT = wpk.shape[:2]
    dev = wpk.device
    w = torch.zeros((NT * 16, KG * 16), dtype=wpk.dtype, device=dev)
    kg = torch.arange(KG, device=dev)[:, None, None, None]
    nt = torch.arange(NT, device=dev)[None, :, None, None]
    li = _LI.to(dev)[None, None, :, None]
    lq = _LQ.to(dev)[None, None, :, None]
    s = torch.arange(4, device=dev)[None, None, None, :]
    w[(nt * 16 + li).expand_as(wpk), (kg * 16 + lq * 4 + s).expand_as(wpk)] = wpk
    return w[:N]


def gemm_split_gather(w):
    """w [N, K] (N % 16 == 0, K % 32 == 0, any dtype) -> [K/32][N/16][64][8] in v_mfma_f32_16x16x32_f16 A-fragment order:
    lane (i = lane & 15, q = lane >> 4) element el = w[16 nt + i][32 ks + 8 q + el]."""
    N, K = w.shape
    assert N % 16 == 0 and K % 32 == 0, (N, K)
    dev = w.device
    ks = torch.arange(K // 32, device=dev)[:, None, None, None]
    nt = torch.arange(N // 16, device=dev)[None, :, None, None]
    li = _LI.to(dev)[None, None, :, None]
    lq = _LQ.to(dev)[None, None, :, None]
    el = torch.arange(8, device=dev)[None, None, None, :]
    return w[nt * 16 + li, ks * 32 + lq * 8 + el].contiguous()


def pack_gemm_split(w):
    """w [N, K] -> (int16 [K/32][N/16][2][64][8], e): the two f16 pieces of the weights (split_f16) for csrc/gemm_split.hip; the same
    number of bytes as pack_gemm(w)"""
    w1, w2, e = split_f16(gemm_split_gather(w))
    return torch.stack([w1, w2], 2).contiguous().view(torch.int16), e


def pack_conv3x3(w, cc, perm=None):
    """w [Cout, Cin, 3, 3] -> [Cin/cc * 9 * cc/16 + 1][CT][64][4]; the trailing zero step pads the prefetch.
    perm: optional list mapping kernel channel slot -> canonical output channel (or -1 for an empty slot)."""
    Cout, Cin = w.shape[:2]
    assert Cin % cc == 0 and cc % 16 == 0
    dev = w.device
    if perm is not None:
        perm_t = torch.as_tensor(perm, device=dev)
        wk = torch.zeros((len(perm), Cin, 3, 3), dtype=w.dtype, device=dev)
        valid = perm_t >= 0
        wk[valid] = w[perm_t[valid]]
        w = wk
    CT = (w.shape[0] + 15) // 16
    w = _pad_rows(w, CT * 16).reshape(CT * 16, Cin, 9)
    nch, ncg = Cin // cc, cc // 16
    chunk = torch.arange(nch, device=dev)[:, None, None, None, None, None]
    tap = torch.arange(9, device=dev)[None, :, None, None, None, None]
    cgl = torch.arange(ncg, device=dev)[None, None, :, None, None, None]
    ct = torch.arange(CT, device=dev)[None, None, None, :, None, None]
    li = _LI.to(dev)[None, None, None, None, :, None]
    lq = _LQ.to(dev)[None, None, None, None, :, None]
    s = torch.arange(4, device=dev)[None, None, None, None, None, :]
    out = w[ct * 16 + li, chunk * cc + cgl * 16 + lq * 4 + s, tap]          # [nch, 9, ncg, CT, 64, 4]
    out = out.reshape(nch * 9 * ncg, CT, 64, 4)
    pad = torch.zeros((1, CT, 64, 4), dtype=w.dtype, device=dev)
    return torch.cat([out, pad], 0).contiguous()


def pack_conv4x4(w):
    """w [Cout, Cin, 4, 4] (Cin % 16 == 0, Cout % 16 == 0) -> [16 taps][Cin/16][CT][64][4]."""
    Cout, Cin = w.shape[:2]
    assert Cin % 16 == 0 and Cout % 16 == 0
    dev = w.device
    CT = Cout // 16
    w = w.reshape(Cout, Cin, 16)
    tap = torch.arange(16, device=dev)[:, None, None, None, None]
    cg = torch.arange(Cin // 16, device=dev)[None, :, None, None, None]
    ct = torch.arange(CT, device=dev)[None, None, :, None, None]
    li = _LI.to(dev)[None, None, None, :, None]
    lq = _LQ.to(dev)[None, None, None, :, None]
    s = torch.arange(4, device=dev)[None, None, None, None, :]
    return w[ct * 16 + li, cg * 16 + lq * 4 + s, tap].contiguous()


def conv4x4_split_gather(w):
    """w [Cout, Cin, 4, 4] (Cin = 16 or a multiple of 32, Cout % 16 == 0, any dtype) -> [NSTEP][CT][64][8] in
    v_mfma_f32_16x16x32_f16 A-fragment order for conv4x4s2_split_kernel (csrc/conv_enc_split.hip): k-step s of lane
    (i = lane & 15, q = lane >> 4) holds tap / 8-channel group
        Cin = 16: 2 s + (q >> 1) / q & 1       Cin = 32: s / q       Cin = 64: s >> 1 / 4 (s & 1) + q     (Cin = 32 m: s // m / 4 (s % m) + q)
    element el = w[16 ct + i][8 group + el][tap]."""
    Cout, Cin = w.shape[:2]
    assert (Cin == 16 or Cin % 32 == 0) and Cout % 16 == 0
    dev = w.device
    CT = Cout // 16
    w = w.reshape(Cout, Cin, 16)
    nstep = 8 if Cin == 16 else 16 * (Cin // 32)
    st = torch.arange(nstep, device=dev)[:, None, None, None]
    ct = torch.arange(CT, device=dev)[None, :, None, None]
    li = _LI.to(dev)[None, None, :, None]
    lq = _LQ.to(dev)[None, None, :, None]
    el = torch.arange(8, device=dev)[None, None, None, :]
    if Cin == 16:
        tap, grp = 2 * st + lq // 2, lq % 2
    else:
        m = Cin // 32
        tap, grp = st // m, 4 * (st % m) + lq
    return w[ct * 16 + li, grp * 8 + el, tap].contiguous()


def pack_conv4x4_split(w):
    """-> (int16 [NSTEP][CT][2][64][8], e): the two f16 pieces in the layout of conv4x4_split_gather"""
    w1, w2, e = split_f16(conv4x4_split_gather(w))
    return torch.stack([w1, w2], 2).contiguous().view(torch.int16), e


def conv4x4_split_index(shape, offset):
    n = 1
    for d in shape:
        n *= d
    ids = (torch.arange(n, dtype=torch.float64) + (offset + 1)).view(shape)
    return (conv4x4_split_gather(ids).reshape(-1) - 1).to(torch.int32)


def pack_conv4x4_image(w):
    """First encoder layer, w [16, 3, 4, 4] -> [3*4 (ci, ky)][CT=1][64]: lane (i, kk) holds w[i][ci][ky][kx = kk]."""
    Cout, Cin = w.shape[:2]
    assert Cin == 3 and Cout % 16 == 0
    dev = w.device
    CT = Cout // 16
    ci = torch.arange(3, device=dev)[:, None, None, None]
    ky = torch.arange(4, device=dev)[None, :, None, None]
    ct = torch.arange(CT, device=dev)[None, None, :, None]
    li = _LI.to(dev)[None, None, None, :]
    lq = _LQ.to(dev)[None, None, None, :]
    return w[ct * 16 + li, ci, ky, lq].reshape(12, CT, 64).contiguous()


def pad_vec(v, n):
    if v.shape[0] == n:
        return v.contiguous()
    out = torch.zeros(n, dtype=v.dtype, device=v.device)
    out[:v.shape[0]] = v
    return out


# ---- discrete-logistic-mixture head: kernel channel order ------------------------------------------------
def dlm_channel_perm(n_mix=10):
    """Kernel slot -> canonical head channel (PixelCNN++ order: [logits(nm) | per colour c: means(nm),
    log_scales(nm), coeffs(nm)]), -1 for an empty slot.  100 real channels in slots 0..99, memory pitch 112:
      slots 8k..8k+7 = {logit_k, mean_r, mean_g, mean_b, coeff0, coeff1, coeff2, log_scale_r} of mixture k (k < 10),
      slots 80..99 = log_scale_g / log_scale_b of every mixture (dlm_log_scale_slot), slots 100..111 empty.
    Everything the mixture MEAN needs for mixture k sits in lanes (q, q+1) of one 16-channel MFMA tile (csrc/conv3x3.hip
    epilogue); the channels fill 6 full MFMA tiles + a 4-channel remainder (slots 96..99) with no padding inside.
    The g / b log-scales of mixture k = 2 ct + h are placed so that the lane that evaluates mixture k of a pixel in the head
    kernel's epilogue (lane group q with q >> 1 = h) finds them in ITS registers of channel tile 5 after the same row swap as the
    other parameters (csrc/conv3x3_head_split.hip, fused likelihood): tile 5 lane group q' = 2 h + (ct >> 1) holds
    {ls_g, ls_b} of ct & 1 = 0 in registers 0, 1 and of ct & 1 = 1 in registers 2, 3; mixtures 8, 9 are slots 96..99."""
    nm = n_mix
    assert nm == 10, "kernel epilogue is written for 10 mixtures (5 MFMA tiles x 2 mixtures)"
    base = lambda c: nm + c * 3 * nm
    perm = []
    for k in range(nm):
        perm += [k, base(0) + k, base(1) + k, base(2) + k, base(0) + 2 * nm + k, base(1) + 2 * nm + k, base(2) + 2 * nm + k,
                 base(0) + nm + k]
    tail = [-1] * 20
    for c in (1, 2):
        for k in range(nm):
            tail[dlm_log_scale_slot(c, k) - 80] = base(c) + nm + k
    perm += tail
    assert sorted(perm) == list(range(100))
    while len(perm) % 16:
        perm.append(-1)
    return perm


def dlm_log_scale_slot(c, k, n_mix=10):
    """slot of log_scale_{c,k} (the same rule is compiled into csrc/common.h: dlm_ls_slot)"""
    if c == 0:
        return 8 * k + 7
    if k >= 8:
        return 96 + 2 * (k - 8) + (c - 1)
    h, ct = k & 1, k >> 1
    return 80 + 4 * (2 * h + (ct >> 1)) + 2 * (ct & 1) + (c - 1)


def pack_dlm_head(w, perm):
    """Output-head weights w [100, 16, 3, 3] for conv3x3_head_kernel<6, REM>: [9 taps + 1][7][64][4].  Tiles 0..5 are the
    ordinary fragment packs of slots 0..95; the storage of tile 6 holds the 4-channel remainder (slots 96..99) for the
    4x4x1 MFMA: float4 index kg * 4 + c = W[slot 96 + c][ci = 4 kg .. 4 kg + 3] (indices >= 16 unused)."""
    full = pack_conv3x3(w, 16, perm=perm)                      # [10, 7, 64, 4]; tile 6 = slots 96..111 in fragment order
    assert full.shape[1] == 7 and w.shape[1] == 16
    dev = w.device
    perm_t = torch.as_tensor(perm, device=dev)
    rem = w[perm_t[96:100]].reshape(4, 16, 9)                  # [c, ci, tap]
    idx = torch.arange(16, device=dev)
    kg, c = idx // 4, idx % 4
    s = torch.arange(4, device=dev)
    tile = rem[c[:, None], (4 * kg)[:, None] + s[None, :], :]  # [16, 4, 9]
    out = full.clone()
    out[:9, 6] = 0
    out[:9, 6, :16] = tile.permute(2, 0, 1)
    return out.contiguous()


def split_f16(w):
    """w (f32, any shape) -> (w1, w2, e): w 2^e ~= w1 + w2 with f16 pieces, |w 2^e - w1 - w2| <= 2^-24 |w 2^e|.  e puts the largest
    |w| in [2^14, 2^15): below the f16 maximum, and the second piece of every weight within 2^-12 of the largest is a normal f16."""
    amax = float(w.abs().max())
    e = 0 if amax == 0.0 else 14 - int(torch.floor(torch.log2(torch.tensor(amax, dtype=torch.float64))))
    assert -20 <= e <= 100, "split_f16: weight magnitude outside the range the kernels' scale bookkeeping covers"
    ws = w.double() * (2.0 ** e)
    w1 = ws.float().half()                                  # round to nearest even (the scaling is exact)
    w2 = (ws - w1.double()).float().half()
    return w1, w2, e


def conv3x3_split_gather(w, perm=None):
    """w [Cout, 16 n, 3, 3] (any dtype) -> [n][5][CT][64][8] in v_mfma_f32_16x16x32_f16 A-fragment order (pure index shuffling, zeros
    for the 10th tap and the padded channel slots).  Per 16-channel input chunk, k-step s holds taps 2 s and 2 s + 1: lane
    (i = lane & 15, q = lane >> 4) element el = W[16 ct + i][chunk 16 + 8 (q & 1) + el][tap 2 s + (q >> 1)]."""
    Cout, Cin = w.shape[:2]
    assert Cin % 16 == 0
    dev = w.device
    if perm is not None:
        perm_t = torch.as_tensor(perm, device=dev)
        wk = torch.zeros((len(perm), Cin, 3, 3), dtype=w.dtype, device=dev)
        valid = perm_t >= 0
        wk[valid] = w[perm_t[valid]]
        w = wk
    CT = (w.shape[0] + 15) // 16
    w = _pad_rows(w, CT * 16).reshape(CT * 16, Cin, 9)
    w = torch.cat([w, torch.zeros((CT * 16, Cin, 1), dtype=w.dtype, device=dev)], 2)      # tap 9 = 0
    ch = torch.arange(Cin // 16, device=dev)[:, None, None, None, None]
    s = torch.arange(5, device=dev)[None, :, None, None, None]
    ct = torch.arange(CT, device=dev)[None, None, :, None, None]
    li = _LI.to(dev)[None, None, None, :, None]
    lq = _LQ.to(dev)[None, None, None, :, None]
    el = torch.arange(8, device=dev)[None, None, None, None, :]
    return w[ct * 16 + li, ch * 16 + (lq % 2) * 8 + el, 2 * s + lq // 2].contiguous()        # [n, 5, CT, 64, 8]


def pack_conv3x3_split(w, perm=None):
    """w [Cout, 16 n, 3, 3] -> (int16 [n][5][CT][2][64][8], e): the two f16 pieces of the weights (split_f16) in the layout of
    conv3x3_split_gather, for csrc/conv3x3_split.hip.  The device-side equivalent is gcpx_split_pack over conv3x3_split_index."""
    g = conv3x3_split_gather(w, perm)
    w1, w2, e = split_f16(g)
    return torch.stack([w1, w2], 3).contiguous().view(torch.int16), e


def conv3x3_split_index(shape, offset, perm=None):
    """int32 [n * 5 * CT * 512]: index into the flat parameter vector of every element of conv3x3_split_gather for a weight of
    `shape` stored at `offset` (-1 = zero)."""
    n = 1
    for d in shape:
        n *= d
    ids = (torch.arange(n, dtype=torch.float64) + (offset + 1)).view(shape)
    return (conv3x3_split_gather(ids, perm).reshape(-1) - 1).to(torch.int32)


def conv3x3_split32_gather(w):
    """w [16 CT, 32 n, 3, 3] -> [n][9 taps][CT][64][8] for conv3x3_up32_split_kernel: one k-step = (tap, 32-channel chunk);
    lane (i = lane & 15, q = lane >> 4) element el = W[16 ct + i][32 chunk + 8 q + el][tap]."""
    Cout, Cin = w.shape[:2]
    assert Cin % 32 == 0 and Cout % 16 == 0
    dev = w.device
    CT = Cout // 16
    w = w.reshape(Cout, Cin, 9)
    ch = torch.arange(Cin // 32, device=dev)[:, None, None, None, None]
    tap = torch.arange(9, device=dev)[None, :, None, None, None]
    ct = torch.arange(CT, device=dev)[None, None, :, None, None]
    li = _LI.to(dev)[None, None, None, :, None]
    lq = _LQ.to(dev)[None, None, None, :, None]
    el = torch.arange(8, device=dev)[None, None, None, None, :]
    return w[ct * 16 + li, ch * 32 + lq * 8 + el, tap].contiguous()


def pack_conv3x3_split32(w):
    """-> (int16 [n][9][CT][2][64][8], e): the two f16 pieces in the layout of conv3x3_split32_gather"""
    w1, w2, e = split_f16(conv3x3_split32_gather(w))
    return torch.stack([w1, w2], 3).contiguous().view(torch.int16), e


def conv3x3_split32_index(shape, offset):
    n = 1
    for d in shape:
        n *= d
    ids = (torch.arange(n, dtype=torch.float64) + (offset + 1)).view(shape)
    return (conv3x3_split32_gather(ids).reshape(-1) - 1).to(torch.int32)


# ---- output head in 32x32x16 MFMA tiles (csrc/conv3x3_head32.hip) ----------------------------------------------------------------
def head32_slot(c, i, n_mix=10):
    """Row i (0..31) of 32-channel tile c (0..3) of the 32x32 head kernel -> slot of the 112-slot layout (dlm_channel_perm), -1 = empty.
    v_mfma_f32_32x32x16_f16 leaves lane l (half h = l >> 5) with rows i = 8 g + 4 h + r (g, r = 0..3) of every tile, all of ONE pixel: the
    lane evaluates mixtures k = 2 m + h (m = 0..4) of that pixel and finds mixture m's eight slots {logit, mean r g b, coeff 0 1 2,
    log_scale_r} in tile m >> 1 at registers 8 (m & 1) + p, and its green / blue log-scales in tile 3 at registers 2 m + (c - 1) —
    no lane exchange.  The same rule is compiled into the kernel (head32_slot)."""
    g, h, r = i // 8, (i // 4) % 2, i % 4
    if c < 3:
        m, p = 2 * c + g // 2, 4 * (g % 2) + r
        return 8 * (2 * m + h) + p if m < 5 else -1
    u = 4 * g + r
    return dlm_log_scale_slot(1 + u % 2, 2 * (u // 2) + h, n_mix) if u < 10 else -1


def head32_gather(w, perm):
    """w [100, 16, 3, 3] -> [9 taps][4 tiles][64][8] in v_mfma_f32_32x32x16_f16 A-fragment order: lane l (i = l & 31, kh = l >> 5)
    element el = W[channel of head32_slot(tile, i)][8 kh + el][tap] (zeros for empty slots)"""
    assert w.shape[1] == 16 and w.shape[0] == 100
    dev = w.device
    chan = torch.tensor([[perm[head32_slot(c, i)] if head32_slot(c, i) >= 0 else -1 for i in range(32)] for c in range(4)], device=dev)
    wz = torch.cat([w.reshape(100, 16, 9), torch.zeros((1, 16, 9), dtype=w.dtype, device=dev)], 0)       # row 100 = zeros
    chan = torch.where(chan >= 0, chan, torch.full_like(chan, 100))
    tap = torch.arange(9, device=dev)[:, None, None, None]
    ct = torch.arange(4, device=dev)[None, :, None, None]
    lane = torch.arange(64, device=dev)[None, None, :, None]
    el = torch.arange(8, device=dev)[None, None, None, :]
    return wz[chan[ct, lane % 32], (lane // 32) * 8 + el, tap].contiguous()                                 # [9, 4, 64, 8]


def pack_head32_split(w, perm):
    """-> (int16 [9][4][2][64][8], e): the two f16 pieces in the layout of head32_gather (host twin of gcpx_split_pack over head32_index)"""
    w1, w2, e = split_f16(head32_gather(w, perm))
    return torch.stack([w1, w2], 2).contiguous().view(torch.int16), e


def head32_index(shape, offset, perm):
    n = 1
    for d in shape:
        n *= d
    ids = (torch.arange(n, dtype=torch.float64) + (offset + 1)).view(shape)
    return (head32_gather(ids, perm).reshape(-1) - 1).to(torch.int32)


_FOLD_CF = (((0.75, 0.25, 0.0), (0.25, 0.75, 0.75), (0.0, 0.0, 0.25)),
            ((0.25, 0.0, 0.0), (0.75, 0.75, 0.25), (0.0, 0.25, 0.75)))


def fold_up_weights(w):
    """w [Cout, Cin, 3, 3] -> f32 [24, Cout, Cin]: the row-folded weights of an upsampling block (bilinear x2, align_corners=False, then
    the 3x3 conv): set py 9 + dyl 3 + tx = sum_ty cf[py][dyl][ty] W[:, :, ty, tx] (output row 2 y + py reads low-resolution row
    y - 1 + dyl), sets 18 + tx = -W[:, :, 0, tx] and 21 + tx = -W[:, :, 2, tx] (the conv's zero rows above / below the image).  Sums in
    float64 in tap-row order, as gcpx_fold_upsample_weights forms them (bit-identical)."""
    wd = w.double()
    out = []
    for py in range(2):
        for dyl in range(3):
            for tx in range(3):
                c = _FOLD_CF[py][dyl]
                s = c[0] * wd[:, :, 0, tx]
                s = s + c[1] * wd[:, :, 1, tx]
                s = s + c[2] * wd[:, :, 2, tx]
                out.append(s)
    out += [-wd[:, :, 0, tx] for tx in range(3)] + [-wd[:, :, 2, tx] for tx in range(3)]
    return torch.stack(out, 0).float()


def conv3x3_fold_gather(fw):
    """fw [24, 16, 32] (fold_up_weights of a 32 -> 16 channel block, any dtype) -> [24][64][8] in v_mfma_f32_16x16x32_f16 A-fragment
    order: lane (i = lane & 15, q = lane >> 4) element el = fw[t][i][8 q + el]."""
    assert fw.shape == (24, 16, 32)
    dev = fw.device
    t = torch.arange(24, device=dev)[:, None, None]
    li = _LI.to(dev)[None, :, None]
    lq = _LQ.to(dev)[None, :, None]
    el = torch.arange(8, device=dev)[None, None, :]
    return fw[t, li, lq * 8 + el].contiguous()


def pack_conv3x3_fold(w):
    """w [16, 32, 3, 3] -> (int16 [24][2][64][8], e): the two f16 pieces of the row-folded weights for conv3x3_up16_fold_kernel
    (csrc/conv3x3_split.hip, GCPX_SPLIT_ROWFOLD).  Device-side: gcpx_fold_upsample_weights + gcpx_split_pack over conv3x3_fold_index."""
    w1, w2, e = split_f16(conv3x3_fold_gather(fold_up_weights(w)))
    return torch.stack([w1, w2], 1).contiguous().view(torch.int16), e


def conv3x3_fold_index():
    """int32 [24 * 512]: index into the flat [24, 16, 32] folded-weight scratch of every element of conv3x3_fold_gather"""
    ids = torch.arange(24 * 16 * 32, dtype=torch.float64).view(24, 16, 32)
    return conv3x3_fold_gather(ids).reshape(-1).to(torch.int32)


def conv3x3_fold16_gather(fw, cbase=0):
    """fw [24, 16, Cin] (fold_up_weights, any dtype), 16 of its input channels from `cbase` -> [16][64][8] for conv3x3_up16_fold16_kernel
    (GCPX_SPLIT_ROWFOLD16): a k-step holds two horizontal taps x 16 channels.  Fragment py 6 + dyl 2 + kind (12 of them), then the border
    corrections 12 + kind (rows above the image) and 14 + kind (below): lane (i = lane & 15, q = lane >> 4) element el =
    fw[set][i][cbase + 8 (q & 1) + el] with set = base + (q >> 1) for kind 0 (taps tx 0 | 1) and base + 2 for kind 1 (tap tx 2; lane
    groups q >> 1 = 1 hold zeros), base = py 9 + dyl 3, 18 or 21."""
    assert fw.shape[0] == 24 and fw.shape[1] == 16 and fw.shape[2] >= cbase + 16
    dev = fw.device
    bases = [py * 9 + dyl * 3 for py in range(2) for dyl in range(3)] + [18, 21]
    fz = torch.cat([fw, torch.zeros((1,) + tuple(fw.shape[1:]), dtype=fw.dtype, device=dev)], 0)      # set 24 = zeros
    li, lq = _LI.to(dev)[:, None], _LQ.to(dev)[:, None]
    el = torch.arange(8, device=dev)[None, :]
    frags = []
    for b in bases:
        for kind in range(2):
            tset = (b + lq // 2) if kind == 0 else torch.where(lq // 2 == 0, torch.full_like(lq, b + 2), torch.full_like(lq, 24))
            frags.append(fz[tset, li, cbase + 8 * (lq % 2) + el])
    # order: [py][dyl][kind] x 12, then [top, bottom][kind]
    return torch.stack(frags, 0).contiguous()


def pack_conv3x3_fold16(w, cbase=0):
    """w [16, Cin, 3, 3] -> (int16 [16][2][64][8], e): the two f16 pieces of the row-folded weights of input channels cbase .. cbase + 15
    in the two-taps-per-k-step order of conv3x3_fold16_gather"""
    w1, w2, e = split_f16(conv3x3_fold16_gather(fold_up_weights(w), cbase))
    return torch.stack([w1, w2], 1).contiguous().view(torch.int16), e


def conv3x3_fold16_index(cin, cbase):
    """int32 [16 * 512]: index into the flat [24, 16, cin] folded-weight scratch of every element of conv3x3_fold16_gather (-1 = zero)"""
    ids = (torch.arange(24 * 16 * cin, dtype=torch.float64) + 1).view(24, 16, cin)
    return (conv3x3_fold16_gather(ids, cbase).reshape(-1) - 1).to(torch.int32)


def lstm_gate_interleave(w_ih, w_hh, b_ih, b_hh):
    """[4H, H] x2 (torch gate order i, f, g, o) -> W [4H, 2H] with row n = 4u + gate, bias [4H] likewise."""
    H = w_hh.shape[1]                                       # (w_ih may be wider than H: embedding folded into layer 0)
    w = torch.cat([w_ih, w_hh], dim=1)                      # [4H, K], K = [x | h]
    K = w.shape[1]
    w = w.view(4, H, K).permute(1, 0, 2).reshape(4 * H, K)
    b = (b_ih + b_hh).view(4, H).t().reshape(4 * H)
    return w.contiguous(), b.contiguous()
