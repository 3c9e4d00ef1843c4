"""Evaluation harness on the device: alignment of predictions to ground truth, image metrics, top-of-100 evaluation.

Counterparts (paths relative to /root/reference):
  DTWEvalBinding / BalancedEvalBinding / BalancedPrunedDTWBinding     gcp/evaluation/evaluation_matching.py:123-221
  TreeDenseRec.get_sample_with_len / get_all_samples_with_len          gcp/prediction/models/tree/tree_dense_rec.py:13-40
  Evaluator.eval / eval_single / compute_metrics / _get_best_idxs      gcp/evaluation/compute_metrics.py:49-141,236-250
The reference walks the batch on the host (numpy / Cython DTW per sequence, skimage metrics per frame); here a whole batch is
aligned and scored by a handful of launches: image cost matrix (gcpx_cdist, f32 MFMA), DTW wavefront + traceback
(gcpx_dtw_align, float64), per-frame squared error + SSIM map (gcpx_image_metrics).

Metric spec (blox.torch.evaluation is absent — this build's written spec, restated in oracle/metrics_oracle.py): mse over the
[-1, 1] images; psnr = mean over frames of 10 log10(1 / mse_frame) on the [0, 1]-scaled images; ssim = mean over frames and
channels of the 7x7 uniform-window SSIM map (K1 = .01, K2 = .03, data range 1, sample covariance: skimage's defaults).

The alignment below is the counterpart of the reference's
`DTWEvalBinding` (/root/reference/gcp/evaluation/evaluation_matching.py:123-146), which `TreeDenseRec.get_sample_with_len`
(tree_dense_rec.py:13-30) and `Evaluator.eval_single` (compute_metrics.py:89-120) call per sequence on the host with
numpy / Cython DTW (dtw_utils.py:77-115, cutils.pyx).  Here the whole batch is aligned by three launches: image cost
matrix (gcpx_cdist, f32 MFMA), DTW wavefront + traceback (gcpx_dtw_align, float64), gather of the chosen frames."""
import torch

from . import runtime as rt


class DTWEvalBinding:
    def __init__(self, model):
        self.m = model
        self.lib = model.lib

    def __call__(self, outputs, inputs, length, i_ex, targets=None, estimates=None):
        seqs, info = self.get_all_samples(outputs, inputs)
        return seqs[i_ex], info

    def get_all_samples(self, outputs, inputs, estimates=None, est_len=None):
        """estimates: [B, N, 3, H, W] (default: every tree node, depth-first = `_collect_sequence`); targets: traj_seq up to
        end_ind.  Returns (list of gen_images [len_b, 3, H, W], Outputs(inds, dist, path, path_len, acc))."""
        m, lib = self.m, self.lib
        est = (outputs.images_df if estimates is None else estimates).contiguous()
        tgt = inputs["traj_seq"].contiguous()
        B, N = est.shape[:2]
        T = tgt.shape[1]
        D = est[0, 0].numel()
        dev = est.device
        st = torch.cuda.current_stream(dev).cuda_stream
        ns = lib.gcpx_cdist_splits(D)
        part = torch.empty(ns, B, N, T, device=dev)
        xn, yn, dsum = torch.empty(B * N, device=dev), torch.empty(B * T, device=dev), torch.empty(B, N, T, device=dev)
        rt.check(lib.gcpx_cdist(est.data_ptr(), tgt.data_ptr(), B, N, T, D, part.data_ptr(), xn.data_ptr(), yn.data_ptr(),
                                dsum.data_ptr(), st), "cdist")
        cost = dsum / float(D)                                   # cdist(..., reduction='mean') (evaluation_matching.py:135)
        t_len = (inputs["end_ind"] + 1).to(torch.int32)
        acc = torch.empty(B, N, T, dtype=torch.float64, device=dev)
        inds = torch.empty(B, T, dtype=torch.int32, device=dev)
        path = torch.empty(B, 2, N + T, dtype=torch.int32, device=dev)
        plen = torch.empty(B, dtype=torch.int32, device=dev)
        dist = torch.empty(B, dtype=torch.float64, device=dev)
        n_len = est_len.to(torch.int32).contiguous() if est_len is not None else None
        rt.check(lib.gcpx_dtw_align(cost.data_ptr(), rt.ptr(n_len), t_len.data_ptr(), B, N, T, acc.data_ptr(), inds.data_ptr(),
                                    path.data_ptr(), plen.data_ptr(), dist.data_ptr(), st), "dtw_align")
        gen = torch.empty((B, T) + tuple(est.shape[2:]), device=dev)
        rt.check(lib.gcpx_gather_rows(est.data_ptr(), inds.data_ptr(), gen.data_ptr(), B, T, N, 0, D, st), "gather")
        lens = t_len.tolist()
        from .model import Outputs
        return [gen[b, :lens[b]] for b in range(B)], Outputs(inds=inds, dist=dist, path=path, path_len=plen, acc=acc, cost=cost,
                                                             pool=est.reshape(B * N, *est.shape[2:]), pool_rows=N)


def mse_cropped(gen_list, inputs):
    """Evaluator.eval_single / compute_metrics (compute_metrics.py:96-101,123-130): first and last frame are conditioning
    frames and are cropped; mse over what remains.  Returns a list of floats (host), one per sequence."""
    out = []
    for b, g in enumerate(gen_list):
        tgt = inputs["traj_seq"][b, :g.shape[0]]
        out.append(float(((g[1:-1] - tgt[1:-1]) ** 2).mean()) if g.shape[0] > 2 else float("nan"))
    return out


class BalancedEvalBinding:
    """evaluation_matching.py:180-206: the nodes a balanced tree of the sequence's length keeps (c_n_prime non-zero), in temporal
    order — what the forward already leaves in `pruned_padded` / `model_enc_seq_padded`."""

    def __init__(self, model):
        self.m = model

    def get_all_samples(self, outputs, inputs=None, length=None, name=None):
        o = outputs.raw
        lens = o["pruned_len" if "pruned_len" in o else "seq_len"].tolist()
        name = "images" if name is None else name
        if name == "images":
            src = o["pruned_padded"]
        elif name in ("e_g_prime", "encodings"):
            src = o["model_enc_seq_padded"]
        else:
            raise KeyError(name)
        return [src[b, :lens[b]] for b in range(len(lens))], None

    def __call__(self, outputs, inputs, length, i_ex, name=None):
        seqs, _ = self.get_all_samples(outputs, inputs, length, name)
        return seqs[i_ex], None


class BalancedPrunedDTWBinding:
    """evaluation_matching.py:209-221: prune with the balanced binding, then warp the pruned sequence onto the ground truth"""

    def __init__(self, model):
        self.m = model
        self.dtw = DTWEvalBinding(model)

    def get_all_samples(self, outputs, inputs, length=None, name=None):
        o = outputs.raw
        est = o["pruned_padded"]                               # [B, T, 3, H, W], rows >= pruned length are zero
        n_len = o["pruned_len" if "pruned_len" in o else "seq_len"]
        return self.dtw.get_all_samples(outputs, inputs, estimates=est, est_len=n_len)

    def __call__(self, outputs, inputs, length, i_ex, name=None):
        seqs, info = self.get_all_samples(outputs, inputs, length, name)
        return seqs[i_ex], info


def get_eval_binding(model, pruning_scheme):
    """TreeDenseRec._get_eval_binding (tree_dense_rec.py:32-40)"""
    if pruning_scheme == "dtw":
        return DTWEvalBinding(model)
    if pruning_scheme in ("pruned_dtw", "basic"):
        assert model._hp.matching_type == "balanced"
        return BalancedPrunedDTWBinding(model) if pruning_scheme == "pruned_dtw" else BalancedEvalBinding(model)
    raise ValueError("Eval pruning scheme {} not currently supported!".format(pruning_scheme))


def image_metrics(model, est_pool, frame_map, tgt, first, last):
    """(mse, psnr, ssim) per sequence, float32 [B, 3] on the device.  tgt [B,T,C,H,W]; est_pool [R,C,H,W]; frame_map int32 [B,T]
    (pool frame matched to target frame (b, t), negative = none); frames t in [first[b], last[b]) are averaged."""
    B, T, C, H, W = tgt.shape
    dev = tgt.device
    scratch = torch.empty(2 * B * T * C, dtype=torch.float64, device=dev)
    out = torch.empty(B, 3, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    rt.check(model.lib.gcpx_image_metrics(est_pool.contiguous().data_ptr(), tgt.contiguous().data_ptr(), rt.ptr(frame_map),
                                          first.data_ptr(), last.data_ptr(), B, T, C, H, W, scratch.data_ptr(), out.data_ptr(), st),
             "image_metrics")
    return out


class Evaluator:
    """compute_metrics.py:49-141: metrics of the prediction against the ground truth, optionally the best of `top_of`
    prior samples per sequence (`top_of_100_eval`).  Everything per-sample stays on the device; `eval` returns / accumulates
    [B, top_of] arrays of mse / psnr / ssim and the index of the best sample under `top_comp_metric`."""

    LOWER_IS_BETTER_METRICS = ["mse"]
    HIGHER_IS_BETTER_METRICS = ["psnr", "ssim"]

    def __init__(self, model, pruning_scheme="dtw", top_of_100=True, top_of=100, top_comp_metric="mse"):
        self.m = model
        self._binding = get_eval_binding(model, pruning_scheme)
        self._scheme = pruning_scheme
        self._top_of = top_of if top_of_100 else 1
        self._top_comp_metric = top_comp_metric
        self.full_evaluation = None

    def reset(self):
        self.full_evaluation = None

    def eval_single(self, inputs, outputs):
        """metrics [B, 3] of one prediction (compute_metrics.py:89-130): align / prune, crop the two conditioning frames, score"""
        tgt = inputs["traj_seq"]
        B, T = tgt.shape[:2]
        dev = tgt.device
        end = inputs["end_ind"].to(torch.int32)
        first = torch.ones(B, dtype=torch.int32, device=dev)
        if self._scheme == "basic":
            # BalancedEvalBinding returns the kept nodes in temporal order: frame t of the prediction is pruned row t
            o = outputs.raw
            est = o["pruned_padded"]
            n_len = o["pruned_len" if "pruned_len" in o else "seq_len"].to(torch.int32)
            fmap = (torch.arange(B, device=dev, dtype=torch.int32)[:, None] * est.shape[1] +
                    torch.arange(T, device=dev, dtype=torch.int32)[None]).contiguous()
            last = torch.minimum(end, n_len - 1)               # input_seq[1:-1] against gen_seq[1:-1]
            pool = est.reshape(-1, *est.shape[2:])
        else:
            _, info = self._binding.get_all_samples(outputs, inputs)
            pool, N = info.pool, info.pool_rows
            fmap = (torch.arange(B, device=dev, dtype=torch.int32)[:, None] * N + info.inds).to(torch.int32)
            fmap = torch.where(info.inds >= 0, fmap, torch.full_like(fmap, -1)).contiguous()
            last = end                                         # frames 1 .. end_ind - 1
        return image_metrics(self.m, pool, fmap, tgt, first, last)

    def _is_better(self, a, b):
        return a < b if self._top_comp_metric in self.LOWER_IS_BETTER_METRICS else a > b

    def eval(self, inputs, outputs=None, model=None, noises=None):
        """Evaluator.eval (compute_metrics.py:132-141): with top-of-N the model is re-run N times under the caller's val_mode
        (prior samples, train.py:205-210) and every sample is scored; `noises` (optional, [N, B, n_nodes, nz_vae]) fixes the draws.
        Returns dict(mse, psnr, ssim: [B, N] float32 on the host, best: [B] index of the best sample)."""
        model = model or self.m
        cols = []
        for n in range(self._top_of):
            if self._top_of > 1 or outputs is None:
                outputs = model(inputs, "train", noise=None if noises is None else noises[n])
            cols.append(self.eval_single(inputs, outputs))
        vals = torch.stack(cols, 1).cpu()                      # [B, N, 3]: the only host transfer
        res = {"mse": vals[..., 0].numpy(), "psnr": vals[..., 1].numpy(), "ssim": vals[..., 2].numpy()}
        res["best"] = self._get_best_idxs(res[self._top_comp_metric])
        if self.full_evaluation is None:
            self.full_evaluation = {k: v.copy() for k, v in res.items()}
        else:
            for k in res:
                self.full_evaluation[k] = __import__("numpy").concatenate((self.full_evaluation[k], res[k]), 0)
        return res

    def _get_best_idxs(self, vals):
        import numpy as np
        return np.argmin(vals, 1) if self._top_comp_metric in self.LOWER_IS_BETTER_METRICS else np.argmax(vals, 1)

    def dump_metrics(self):
        """compute_metrics.py:214-226: per metric (mean, std of the best samples, mean per-sequence std over samples)"""
        import numpy as np
        out = {}
        best = self.full_evaluation["best"]
        for k in ("mse", "psnr", "ssim"):
            v = self.full_evaluation[k]
            bv = v[np.arange(v.shape[0]), best]
            out[k] = (float(bv.mean()), float(bv.std()), float(v.std(axis=1).mean()))
        return out
