"""Evaluation alignment of predictions to ground truth on the device — counterpart of the reference's
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

    def get_all_samples(self, outputs, inputs, estimates=None):
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
        rt.check(lib.gcpx_dtw_align(cost.data_ptr(), None, t_len.data_ptr(), B, N, T, acc.data_ptr(), inds.data_ptr(),
                                    path.data_ptr(), plen.data_ptr(), dist.data_ptr(), st), "dtw_align")
        gen = torch.empty((B, T) + tuple(est.shape[2:]), device=dev)
        rt.check(lib.gcpx_gather_rows(est.data_ptr(), inds.data_ptr(), gen.data_ptr(), B, T, N, 0, D, st), "gather")
        lens = t_len.tolist()
        from .model import Outputs
        return [gen[b, :lens[b]] for b in range(B)], Outputs(inds=inds, dist=dist, path=path, path_len=plen, acc=acc, cost=cost)


def mse_cropped(gen_list, inputs):
    """Evaluator.eval_single / compute_metrics (compute_metrics.py:96-101,123-130): first and last frame are conditioning
    frames and are cropped; mse over what remains.  Returns a list of floats (host), one per sequence."""
    out = []
    for b, g in enumerate(gen_list):
        tgt = inputs["traj_seq"][b, :g.shape[0]]
        out.append(float(((g[1:-1] - tgt[1:-1]) ** 2).mean()) if g.shape[0] > 2 else float("nan"))
    return out
