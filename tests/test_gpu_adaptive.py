"""-m gpu: the adaptive-binding path (soft-DTW frame binding + attentive inference, config c5) through the C-ABI against
the CPU oracle (oracle/adaptive_oracle.py, itself pinned to goldens produced by executing the reference's soft_dtw) and
directly against those goldens.  Tolerances (fp32; float64 inside the DTW) are written at each comparison."""
import math
import os

import numpy as np
import pytest
import torch

from helpers import make_inputs, assert_close

pytestmark = pytest.mark.gpu

LAT_ATOL, LAT_RTOL = 5e-5, 1e-4
PIX_ATOL = 2e-5
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_dtw.npz"))


def _lib():
    from video_gcp_amd import runtime as rt
    return rt, rt.load_library()


def _st():
    return torch.cuda.current_stream().cuda_stream


@pytest.mark.parametrize("i", range(int(G["sd_n"])))
def test_soft_dtw_kernel_matches_reference_goldens(i):
    """cost -> forward accumulator (float64) and normalised expected edge frequencies, against outputs of the reference's
    fast_gak / soft_dtw (probabilistic_dtw.py).  D = temp = 1 makes the kernel's cost the golden's cost bit for bit."""
    from oracle import adaptive_oracle as A
    rt, lib = _lib()
    cost, end = torch.tensor(G[f"sd{i}_cost"]), torch.tensor(G[f"sd{i}_end"])
    B, N, T = cost.shape
    acc = torch.zeros(2 * B, N, T, dtype=torch.float64, device="cuda")
    w = torch.zeros(B, N, T, device="cuda")
    one, cd, ed = torch.ones(1, device="cuda"), cost.cuda(), end.cuda()
    rt.check(lib.gcpx_soft_dtw(cd.data_ptr(), 1.0, one.data_ptr(), ed.data_ptr(), B, N, T, acc.data_ptr(), w.data_ptr(), _st()),
             "soft_dtw")
    torch.cuda.synchronize()
    fwd, ref = acc[:B].cpu().numpy(), G[f"sd{i}_fwd"]
    assert np.array_equal(np.isinf(fwd), np.isinf(ref))
    fin = ~np.isinf(ref)
    assert np.max(np.abs(fwd[fin] - ref[fin])) < 1e-11
    want = A.normalize(torch.tensor(G[f"sd{i}_w"]), 1)
    assert_close(w, want, 2e-6, 0, "normalised w")
    # frames after end_ind are never matched; every other frame's column is a distribution over nodes
    for b in range(B):
        e = int(end[b])
        assert float(w[b, :, e + 1:].abs().max()) == 0.0 if e + 1 < T else True
        assert torch.allclose(w[b, :, :e + 1].sum(0).cpu(), torch.ones(e + 1), atol=1e-5)


GT = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_dtw_dtemp.npz"))


@pytest.mark.parametrize("i", list(range(int(GT["n"]))) + ["c5"])
def test_soft_dtw_temperature_gradient_kernel(i):
    """gcpx_soft_dtw_dtemp (learn_matching_temp, adaptive.py:19-21, :51): d / d temp of the averaging criterion through
    normalize(soft_dtw(cost / temp)), against torch autograd over the oracle's soft_dtw_autograd — which the CPU suite pins to central
    differences of the reference's forward executed at temp +- h (tests/golden/ref_dtw_dtemp.npz; the reference's own autograd
    gives NaN there).  Tolerance 2e-5 relative (float64 inside, float32 cost and weights)."""
    from oracle import adaptive_oracle as A
    rt, lib = _lib()
    D, coef, ls0 = 3.0, 0.37, 0.2
    if i == "c5":                                                     # c5's lattice (255 nodes x 200 frames), ragged lengths
        g = torch.Generator().manual_seed(5)
        dsum, end, t0 = torch.rand(4, 255, 200, generator=g) * 1.5 * D, torch.tensor([199, 120, 37, 1]), 0.4
    else:
        dsum, end, t0 = torch.tensor(GT[f"c{i}_cost"]) * D, torch.tensor(GT[f"c{i}_end"]), float(GT[f"c{i}_temp"])
    B, N, T = dsum.shape
    pad = (torch.arange(T)[None] <= end[:, None]).float()
    pad[0, 1] = 0.0                                                   # the mask is an input, not derived from end_ind
    temp = torch.full((1,), t0, requires_grad=True)
    ls = torch.full((1,), ls0)
    w = A.normalize(A.soft_dtw_autograd((dsum / D) / temp, end), 1)
    val = 0.5 * dsum * torch.exp(-ls) ** 2 + D * (ls + 0.5 * math.log(2 * math.pi))
    (want,) = torch.autograd.grad(coef * (val * w * pad[:, None]).sum(), temp)
    acc = torch.zeros(2 * B, N, T, dtype=torch.float64, device="cuda")
    tan = torch.full((2 * B, N, T), float("nan"), dtype=torch.float64, device="cuda")
    part = torch.zeros(B, dtype=torch.float64, device="cuda")
    wd = torch.zeros(B, N, T, device="cuda")
    tp, dd, ed, pd_, lsd = temp.detach().cuda(), dsum.cuda(), end.cuda(), pad.cuda(), ls.cuda()
    got = torch.full((1,), 0.25, device="cuda")                       # the kernel ACCUMULATES into the gradient slot
    rt.check(lib.gcpx_soft_dtw(dd.data_ptr(), D, tp.data_ptr(), ed.data_ptr(), B, N, T, acc.data_ptr(), wd.data_ptr(), _st()), "soft_dtw")
    rt.check(lib.gcpx_soft_dtw_dtemp(dd.data_ptr(), D, tp.data_ptr(), ed.data_ptr(), acc.data_ptr(), pd_.data_ptr(), lsd.data_ptr(), coef,
                                     B, N, T, tan.data_ptr(), part.data_ptr(), got.data_ptr(), _st()), "soft_dtw_dtemp")
    torch.cuda.synchronize()
    assert_close(wd, w.detach(), 2e-6, 0, "w")
    g = float(got) - 0.25
    assert abs(float(want)) > 1e-3
    assert abs(g - float(want)) <= 2e-5 * abs(float(want)) + 1e-6, (g, float(want))
    # twice the same launch: bit-identical (fixed summation order)
    got2 = torch.full((1,), 0.25, device="cuda")
    rt.check(lib.gcpx_soft_dtw_dtemp(dd.data_ptr(), D, tp.data_ptr(), ed.data_ptr(), acc.data_ptr(), pd_.data_ptr(), lsd.data_ptr(), coef,
                                     B, N, T, tan.data_ptr(), part.data_ptr(), got2.data_ptr(), _st()), "soft_dtw_dtemp")
    torch.cuda.synchronize()
    assert float(got2) == float(got)


def test_soft_dtw_full_size_properties():
    """c5 size (255 nodes x 200 frames): agreement with the oracle and the structural properties of the alignment posterior"""
    from oracle import adaptive_oracle as A
    rt, lib = _lib()
    g = torch.Generator().manual_seed(0)
    B, N, T = 3, 255, 200
    dsum = torch.rand(B, N, T, generator=g) * 3000.0
    end = torch.tensor([199, 57, 120])
    temp = torch.tensor([0.7])
    D = 12288.0
    acc = torch.zeros(2 * B, N, T, dtype=torch.float64, device="cuda")
    w = torch.zeros(B, N, T, device="cuda")
    dd, td, ed = dsum.cuda(), temp.cuda(), end.cuda()
    rt.check(lib.gcpx_soft_dtw(dd.data_ptr(), D, td.data_ptr(), ed.data_ptr(), B, N, T, acc.data_ptr(), w.data_ptr(), _st()), "soft_dtw")
    torch.cuda.synchronize()
    want = A.normalize(torch.from_numpy(A.soft_dtw(((dsum / D) / temp).numpy(), end.numpy())), 1)
    assert_close(w, want, 2e-6, 1e-5, "w")
    for b in range(B):
        e = int(end[b])
        assert torch.allclose(w[b, :, :e + 1].sum(0).cpu(), torch.ones(e + 1), atol=1e-5)
        if e + 1 < T:
            assert float(w[b, :, e + 1:].abs().max()) == 0.0


@pytest.mark.parametrize("shape", [(2, 15, 12, 3 * 32 * 32), (2, 70, 65, 256), (1, 255, 200, 3 * 64 * 64)])
def test_cdist_kernel(shape):
    """stated tolerance: 2e-6 of the largest squared norm (quadratic expansion in fp32, summation order differs)"""
    from oracle import adaptive_oracle as A
    rt, lib = _lib()
    B, N, T, K = shape
    g = torch.Generator().manual_seed(1)
    x, y = torch.rand(B, N, K, generator=g) * 2 - 1, torch.rand(B, T, K, generator=g) * 2 - 1
    y[0, 0] = x[0, 0]                                          # an exact match: distance clamps at >= 0
    ns = lib.gcpx_cdist_splits(K)
    part = torch.zeros(ns, B, N, T, device="cuda")
    xn, yn, out = torch.zeros(B * N, device="cuda"), torch.zeros(B * T, device="cuda"), torch.zeros(B, N, T, device="cuda")
    xd, yd = x.cuda(), y.cuda()
    rt.check(lib.gcpx_cdist(xd.data_ptr(), yd.data_ptr(), B, N, T, K, part.data_ptr(), xn.data_ptr(), yn.data_ptr(),
                            out.data_ptr(), _st()), "cdist")
    torch.cuda.synchronize()
    want = A.batch_cdist(x, y, "sum")
    scale = float((x ** 2).sum(-1).max())
    assert_close(out, want, 2e-6 * scale, 0, "cdist")
    assert float(out.min()) >= 0.0


@pytest.mark.parametrize("heads", [1, 2])
def test_attention_kernel(heads):
    import video_gcp_amd as V
    from oracle import adaptive_oracle as A
    rt, lib = _lib()
    hp = V.config("c5s", n_attention_heads=heads)
    g = torch.Generator().manual_seed(2)
    B, T, n, dk, nz = 3, 37, 5, hp.nz_attn_key, hp.nz_enc
    M = B * n
    q, k, v = torch.randn(M, dk, generator=g), torch.randn(B, T, dk, generator=g), torch.randn(B, T, nz, generator=g)
    end = torch.tensor([36, 2, 20])
    ident = lambda d: {"weight": torch.eye(d), "bias": torch.zeros(d)}
    sd = {}
    for nm, d in (("q_proj", dk), ("k_proj", dk), ("v_proj", nz), ("out_proj", nz)):
        for kk, vv in ident(d).items():
            sd[f"a.{nm}.{kk}"] = vv
    sd["a.temperature"] = torch.tensor([0.8])
    rep = lambda t: t.repeat_interleave(n, 0)
    want_o, want_a = A.multihead_attention(sd, "a", hp, q, rep(k), rep(v), torch.zeros(M, dtype=torch.long), rep(end))
    out, att = torch.zeros(M, nz, device="cuda"), torch.zeros(M, T, device="cuda")
    qd, kd, vd, ed, td = q.cuda(), k.cuda(), v.cuda(), end.cuda(), sd["a.temperature"].cuda()
    rt.check(lib.gcpx_attention(qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), None, ed.data_ptr(), td.data_ptr(), out.data_ptr(),
                                att.data_ptr(), M, n, T, dk, nz, heads, _st()), "attention")
    torch.cuda.synchronize()
    assert_close(att, want_a, 1e-6, 1e-5, "attention weights")
    assert_close(out, want_o, 1e-5, 1e-5, "attention output")


def test_match_stats_prune_and_averaging_kernels():
    from oracle import adaptive_oracle as A
    from oracle import gcp_model_oracle as O
    rt, lib = _lib()
    g = torch.Generator().manual_seed(3)
    B, L, T, D = 3, 4, 12, 48
    N = 2 ** L - 1
    w = torch.rand(B, N, T, generator=g)
    w[:, :, 9:] = 0                                      # padded frames: all-zero columns -> root
    w[0, 3, 2] = w[0, 7, 2] = 2.0                        # a tie: the first maximum in BREADTH-first order (the root, df 7) wins
    w = w / w.sum(1, keepdim=True).clamp_min(1e-7)
    end = torch.tensor([8, 8, 5])
    f2n, midx = torch.zeros(B, T, dtype=torch.int32, device="cuda"), torch.zeros(B, T, dtype=torch.int32, device="cuda")
    best = torch.zeros(B, N, dtype=torch.int32, device="cuda")
    ent, pn = torch.zeros(B, N, device="cuda"), torch.zeros(B, N, device="cuda")
    wd, ed = w.cuda(), end.cuda()
    rt.check(lib.gcpx_match_stats(wd.data_ptr(), ed.data_ptr(), B, L, T, f2n.data_ptr(), midx.data_ptr(), best.data_ptr(),
                                  ent.data_ptr(), pn.data_ptr(), _st()), "match_stats")
    w_bf = O._bf_of_df(w, L)
    from oracle import tree_index_oracle as TI
    want_f2n = TI.bf2df_perm(L)[w_bf.argmax(1).numpy()]
    assert np.array_equal(f2n.cpu().numpy(), want_f2n)
    want_m = np.where(np.arange(T)[None] <= end.numpy()[:, None], want_f2n, -1)
    assert np.array_equal(midx.cpu().numpy(), want_m)
    assert np.array_equal(best.cpu().numpy(), w.argmax(-1).numpy())
    assert_close(ent, A.safe_entropy(w, -1), 1e-6, 1e-5, "entropy")
    assert_close(pn, w.sum(2).clamp(0, 1), 1e-6, 0, "p_n")
    # learned pruning + BCE targets
    dist = torch.randn(B, N - 1, generator=g)
    leave, kept = torch.zeros(B, N, dtype=torch.int32, device="cuda"), torch.zeros(B, N, dtype=torch.int32, device="cuda")
    cnt, tgt = torch.zeros(B, dtype=torch.int32, device="cuda"), torch.zeros(B, N - 1, dtype=torch.int32, device="cuda")
    dd = dist.cuda()
    rt.check(lib.gcpx_distance_prune(dd.data_ptr(), 0.5, best.data_ptr(), B, N, leave.data_ptr(), kept.data_ptr(),
                                     cnt.data_ptr(), tgt.data_ptr(), _st()), "distance_prune")
    close = torch.cat([torch.zeros(B, 1, dtype=torch.bool), torch.sigmoid(dist) > 0.5], 1)
    assert torch.equal(leave.cpu().bool(), ~close)
    for b in range(B):
        pos = torch.nonzero(~close[b])[:, 0].int()
        assert int(cnt[b]) == len(pos) and torch.equal(kept[b, :len(pos)].cpu(), pos) and bool((kept[b, len(pos):] == -1).all())
    bt = w.argmax(-1)
    assert torch.equal(tgt.cpu().bool(), bt[:, 1:] == bt[:, :-1])
    # averaging loss rows and the soft average
    dsum, ls = torch.rand(B, N, T, generator=g) * 50, torch.tensor([0.3])
    nll = torch.zeros(B, T, device="cuda")
    dsd, lsd = dsum.cuda(), ls.cuda()
    rt.check(lib.gcpx_averaging_nll(dsd.data_ptr(), wd.data_ptr(), lsd.data_ptr(), float(D), B, N, T, nll.data_ptr(),
                                    _st()), "averaging_nll")
    want = ((0.5 * dsum * torch.exp(-ls) ** 2 + D * (ls + 0.5 * math.log(2 * math.pi))) * w).sum(1)
    assert_close(nll, want, 1e-4, 1e-5, "averaging nll")
    x = torch.randn(B, N, D, generator=g)
    avg = torch.zeros(B, T, D, device="cuda")
    xd = x.cuda()
    rt.check(lib.gcpx_soft_average(wd.data_ptr(), xd.data_ptr(), avg.data_ptr(), B, N, T, D, _st()), "soft_average")
    torch.cuda.synchronize()
    assert_close(avg, torch.einsum("bnt,bnd->btd", w, x), 1e-5, 1e-5, "soft average")


def _build(cfg, **over):
    import video_gcp_amd as V
    from video_gcp_amd.model import GCPTreeModel
    hp = V.config(cfg, **over)
    sd = V.init_params(hp, seed=1, randomize_affine=True)
    sd["decoder.log_sigma"].fill_(0.2)
    sd["tree_module.tree_modules.0.binding.temp"].fill_(0.05)       # a sharp posterior: the matching actually selects
    model = GCPTreeModel(hp, params=sd, device="cuda")
    return hp, sd, model


@pytest.mark.parametrize("variant", ["A", "B"])
@pytest.mark.parametrize("graph", [False, True])
def test_adaptive_forward_and_losses_c5s(variant, graph):
    """whole adaptive forward at a small size (L=4, 15 nodes, T=12, 32x32) against the oracle: attentive posterior, decoded
    nodes, cost matrix, matching distribution, learned pruning, auxiliary heads and every loss term."""
    from oracle import gcp_model_oracle as O
    from oracle import tree_index_oracle as TI
    hp, sd, model = _build("c5s")
    model.use_graph = graph
    model.train(True)
    inputs, noise, _ = make_inputs(hp, seed=11, variant=variant)
    ref = O.forward(sd, hp, inputs, noise=noise, training_bn=True)
    ref_losses, ref_total = O.losses(sd, hp, inputs, ref)
    dev_in = {k: v.cuda() for k, v in inputs.items()}
    for _ in range(2):
        out = model(dev_in, "train", noise=noise.cuda())
    torch.cuda.synchronize()
    bf, tree = ref["tree_bf"], out.tree
    assert_close(tree.bf.e_tilde, bf["e_tilde"], LAT_ATOL, LAT_RTOL, "e_tilde")
    assert_close(tree.bf.gamma, bf["gamma"], 1e-6, 1e-4, "gamma")
    for k_dev, k_ref in (("e_g_prime", "e_g_prime"), ("hidden_state", "hidden"), ("z", "z"), ("q_z_mu", "q_z_mu"),
                         ("q_z_log_sigma", "q_z_log_sigma"), ("p_z_mu", "p_z_mu")):
        assert_close(getattr(tree.bf, k_dev), bf[k_ref], LAT_ATOL, LAT_RTOL, k_dev)
    assert_close(tree.bf.images, bf["images"], PIX_ATOL, 0, "images")
    D = 3 * hp.img_sz ** 2
    assert_close(out.raw["cdist_sum"] / D, ref["cost_df"], 2e-6, 1e-5, "cost matrix")
    # the matching distribution: exp of a sum of up to N costs / temp -> relative 2e-3 at temp 0.05, absolute 2e-5
    assert_close(tree.bf.match_dist, ref["match_dist"], 2e-5, 2e-3, "match_dist")
    assert_close(tree.bf.p_n, ref["p_n"], 2e-5, 2e-3, "p_n")
    assert_close(out.entropy, ref["entropy"], 5e-4, 2e-3, "entropy")
    assert_close(out.distance_predictor.distances, ref["distances"], LAT_ATOL, LAT_RTOL, "distances")
    assert np.array_equal(out.raw["frame2node"].cpu().numpy(), TI.bf2df_perm(hp.hierarchy_levels)[ref["matched_idx"].numpy()])
    assert np.array_equal(out.raw["leave"].cpu().numpy().astype(bool), ref["leave_df"].numpy())
    pruned = model.pruned_prediction(out)
    assert [p.shape[0] for p in pruned] == [p.shape[0] for p in ref["pruned_prediction"]]
    for a, b in zip(pruned, ref["pruned_prediction"]):
        assert_close(a, b, PIX_ATOL, 0, "pruned_prediction")
    aux = model.aux_outputs(out)
    assert_close(aux.model_enc_seq, ref["model_enc_seq"], LAT_ATOL, LAT_RTOL, "model_enc_seq")
    assert_close(aux.regressed_state, ref["regressed_state"], LAT_ATOL, LAT_RTOL, "regressed_state")
    assert_close(aux.actions, ref["actions"], LAT_ATOL, LAT_RTOL, "actions")
    assert_close(model.soft_matched_estimates(out), ref["soft_matched_estimates"], 5e-5, 2e-3, "soft_matched_estimates")
    losses = model.loss(dev_in, out)
    total = model.get_total_loss(dev_in, losses)
    for name, (val, wgt) in ref_losses.items():
        got = float(losses[name].value)
        assert abs(got - float(val)) <= 1e-4 * abs(float(val)) + 1e-6, (name, got, float(val))
        assert losses[name].weight == wgt
    assert abs(float(total.value) - float(ref_total)) <= 1e-4 * abs(float(ref_total)) + 1e-6


def test_adaptive_prior_path_c5s():
    """val_mode(): no matching is computed; pruning comes from the distance predictor (tree.py:58, 69-70)"""
    from oracle import gcp_model_oracle as O
    hp, sd, model = _build("c5s")
    model.eval()
    inputs, noise, _ = make_inputs(hp, seed=12, variant="A")
    plan_in = {k: inputs[k] for k in ("I_0", "I_g", "end_ind", "start_ind")}
    ref = O.forward(sd, hp, plan_in, noise=noise, sample_prior=True, training_bn=False)
    with model.val_mode(pred_length=False):
        out = model({k: v.cuda() for k, v in plan_in.items()}, "train", noise=noise.cuda())
    torch.cuda.synchronize()
    assert_close(out.tree.bf.images, ref["tree_bf"]["images"], PIX_ATOL, 0, "images")
    assert_close(out.distance_predictor.distances, ref["distances"], LAT_ATOL, LAT_RTOL, "distances")
    aux = model.aux_outputs(out)
    assert_close(aux.model_enc_seq, ref["model_enc_seq"], LAT_ATOL, LAT_RTOL, "model_enc_seq")
    for a, b in zip(model.pruned_prediction(out), ref["pruned_prediction"]):
        assert_close(a, b, PIX_ATOL, 0, "pruned_prediction")


def test_adaptive_c5_full_size_properties():
    """BASELINE configs[4] shapes per GPU (64x64, T=200, L=8, 255 nodes; batch 2 here): runs, and the binding has the
    properties the domain guarantees whatever the size."""
    hp, sd, model = _build("c5", batch_size=2)
    model.train(True)
    inputs, noise, _ = make_inputs(hp, seed=13, variant="B")
    dev_in = {k: v.cuda() for k, v in inputs.items()}
    out = model(dev_in, "train", noise=noise.cuda())
    losses = model.loss(dev_in, out)
    torch.cuda.synchronize()
    w = out.raw["match_dist_df"]
    assert torch.isfinite(w).all() and torch.isfinite(out.images_df).all()
    for b in range(hp.batch_size):
        e = int(inputs["end_ind"][b])
        assert torch.allclose(w[b, :, :e + 1].sum(0).cpu(), torch.ones(e + 1), atol=1e-4)
        if e + 1 < hp.max_seq_len:
            assert float(w[b, :, e + 1:].abs().max()) == 0.0
        assert int(out.raw["frame2node"][b, 0]) == 0 and int(out.raw["frame2node"][b, e]) == hp.n_nodes - 1   # path end points
    assert all(math.isfinite(float(v.value)) for k, v in losses.items() if k != "_total")


def test_dtw_align_kernel_matches_reference_goldens():
    """accumulated cost, path and normalised distance against outputs of the reference's basic_dtw (dtw_utils.py) executed
    in the build container; chosen frames against the oracle's restatement of get_single_matches"""
    from oracle import adaptive_oracle as A
    rt, lib = _lib()
    for i in range(int(G["bd_n"])):
        C64 = G[f"bd{i}_C"]
        C = torch.tensor(C64.astype(np.float32))
        n, t = C.shape
        cd = C[None].contiguous().cuda()
        acc = torch.zeros(1, n, t, dtype=torch.float64, device="cuda")
        inds = torch.zeros(1, t, dtype=torch.int32, device="cuda")
        path = torch.zeros(1, 2, n + t, dtype=torch.int32, device="cuda")
        plen, dist = torch.zeros(1, dtype=torch.int32, device="cuda"), torch.zeros(1, dtype=torch.float64, device="cuda")
        rt.check(lib.gcpx_dtw_align(cd.data_ptr(), None, None, 1, n, t, acc.data_ptr(), inds.data_ptr(), path.data_ptr(),
                                    plen.data_ptr(), dist.data_ptr(), _st()), "dtw_align")
        torch.cuda.synchronize()
        d_ref, D_ref, path_ref, inds_ref = A.dtw_matches(C.numpy())       # float32 costs, float64 accumulation
        assert np.max(np.abs(acc[0].cpu().numpy() - D_ref)) < 1e-12
        L = int(plen[0])
        got_p = path[0, :, :L].cpu().numpy()[:, ::-1]
        assert np.array_equal(got_p[0], path_ref[0]) and np.array_equal(got_p[1], path_ref[1])
        assert np.array_equal(inds[0].cpu().numpy(), inds_ref) and abs(float(dist[0]) - d_ref) < 1e-12
        # the float64 goldens of the reference differ from the float32-cost run only by the cost rounding
        assert np.max(np.abs(D_ref - G[f"bd{i}_D"])) < 1e-5 * (n + t)
        assert np.array_equal(path_ref[0], G[f"bd{i}_p0"]) and np.array_equal(path_ref[1], G[f"bd{i}_p1"])


def test_dtw_eval_binding_c5s():
    """DTWEvalBinding over a batch (ragged target lengths) against the oracle applied per sequence"""
    from oracle import adaptive_oracle as A
    from video_gcp_amd.evaluation import DTWEvalBinding, mse_cropped
    hp, sd, model = _build("c5s")
    model.eval()
    inputs, noise, _ = make_inputs(hp, seed=21, variant="B")
    dev_in = {k: v.cuda() for k, v in inputs.items()}
    with model.val_mode(pred_length=False):
        out = model(dev_in, "test", noise=noise.cuda())
    gen, info = DTWEvalBinding(model).get_all_samples(out, dev_in)
    torch.cuda.synchronize()
    est = out.images_df.cpu()
    for b in range(hp.batch_size):
        e = int(inputs["end_ind"][b])
        cost = A.batch_cdist(est[b:b + 1], inputs["traj_seq"][b:b + 1, :e + 1], "mean")[0].numpy()
        d, D, path, inds = A.dtw_matches(cost)
        got_cost = info.cost[b, :, :e + 1].cpu().numpy()
        # the decision variables are discrete: compare them only where the cost matrices agree to rounding
        assert np.max(np.abs(got_cost - cost)) < 2e-6
        assert np.array_equal(info.inds[b, :e + 1].cpu().numpy(), inds)
        assert gen[b].shape[0] == e + 1
        assert_close(gen[b], est[b][inds], 0, 0, "gen_images")
    m = mse_cropped(gen, dev_in)
    assert len(m) == hp.batch_size and all(np.isfinite(x) or inputs["end_ind"][i] < 2 for i, x in enumerate(m))


def _compare_grads(gref, got, rtol=1e-3, atol=5e-7):
    bad = []
    for k, g in gref.items():
        h = got[k].cpu()
        err, scale = float((h - g).abs().max()), float(g.abs().max())
        if err > rtol * scale + atol:
            bad.append((k, err, scale))
    assert not bad, bad[:12]


@pytest.mark.parametrize("learn_temp", [False, True])
@pytest.mark.parametrize("graph", [False, True])
def test_adaptive_gradients_match_autograd_c5s(graph, learn_temp):
    """training step of the adaptive model (explicit backward through the averaging loss, the mixture mean, the attentive
    posterior and both temporal encoders) against torch autograd over the oracle.  Stated tolerance: 1e-3 of each gradient's
    max-abs (+5e-7), as for the balanced model (tests/test_gpu_training.py).  learn_temp: hyperparameters.py:132's default — the
    matching temperature receives the criterion's gradient through the matching weights (zero otherwise)."""
    from oracle import gcp_model_oracle as O
    from video_gcp_amd.training import GCPTrainStep
    hp, sd, model = _build("c5s", learn_matching_temp=learn_temp)
    # a sharp posterior where the temperature is learned: its gradient is ~1e-3 there (8e-6 at 0.3, below the absolute tolerance)
    sd["tree_module.tree_modules.0.binding.temp"].fill_(0.02 if learn_temp else 0.3)
    model.load_state_dict({"tree_module.tree_modules.0.binding.temp": sd["tree_module.tree_modules.0.binding.temp"]}, strict=False)
    model.use_graph = graph
    tr = GCPTrainStep(model, lr=1e-3)
    inputs, noise, _ = make_inputs(hp, seed=31, variant="B")
    dev_in = {k: v.cuda() for k, v in inputs.items()}
    for _ in range(2):
        out = tr.backward(dev_in, noise.cuda())
    torch.cuda.synchronize()
    gref, res, total, _ = O.gradients(sd, hp, inputs, noise)
    assert abs(float(out.raw["losses"][5]) - float(total.detach())) <= 1e-4 * abs(float(total.detach()))
    got = tr.named_grads()
    _compare_grads(gref, got)
    gt = float(got["tree_module.tree_modules.0.binding.temp"].abs().max())
    assert (gt > 1e-4) if learn_temp else (gt == 0.0)


def test_adaptive_training_step_c5_shapes_decreases_loss():
    """configs[4] shapes (64x64, T=200, L=8; batch 2): finite gradients, the loss goes down on a fixed batch"""
    from video_gcp_amd.training import GCPTrainStep
    hp, sd, model = _build("c5", batch_size=2)
    tr = GCPTrainStep(model, lr=2e-3)
    inputs, noise, _ = make_inputs(hp, seed=32, variant="B")
    dev_in = {k: v.cuda() for k, v in inputs.items()}
    losses = []
    for _ in range(5):
        out = tr.step(dev_in, noise.cuda())
        losses.append(float(out.raw["losses"][5]))
    assert all(math.isfinite(x) for x in losses) and bool(torch.isfinite(tr.grad).all())
    assert losses[-1] < losses[0], losses
