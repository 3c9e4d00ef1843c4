"""Sub-module handles of the model object, under the attribute names the reference's callers use:

  model.encoder(img)[0][:, :, 0, 0]                        planner_policy.py:225 (closed-loop execution)
  model.inv_mdl.run_single(enc0, enc1)                     planner_policy.py:226-227, inverse_mdl.py:222-225
  model.cost_mdl.cost_pred(enc1, enc2)                     cost_mdl.py:138-145 (TestTimeCostModel.forward)
  model.decoder.decode_seq(inputs, enc)                    tree_dense_rec.py:42, sequential.py:56
  model.dense_rec.get_sample_with_len / get_all_samples_with_len / eval_binding     tree_dense_rec.py:13-40
  model.step()                                             train.py:163

They are thin: each one records a short launch plan over the same HIP kernels the forward uses (no torch compute)."""
import torch

from . import runtime as rt


class Skips:
    """the encoder's skip activations of one batch of frames: raw (pre-BatchNorm) NHWC tensors + the folded affine the
    consumer applies on load — handed back to `decoder.decode_seq` as `inputs['skips']`"""

    def __init__(self, srcs, n_frames):
        self.srcs, self.n_frames = srcs, n_frames


class EncoderHandle:
    def __init__(self, model):
        self.m = model

    def __call__(self, images):
        """Encoder.forward contract (base_gcp.py:188,208-209): NCHW images in [-1, 1] -> (enc [F, nz_enc, 1, 1], skips)"""
        lat, skips = self.m._encode(images, keep_skips=True)
        return lat[:, :, None, None], skips


class DecoderHandle:
    def __init__(self, model):
        self.m = model

    def nll(self, estimates, targets, weights=1, log_error_arr=False):
        """DecoderModule.nll(estimates, targets, weights) as BalancedBinding.reconstruction_loss calls it (frame_binding.py:88-99):
        `estimates` = the matched distribution parameters in the head kernel's slot order ([B, T, H, W, 112] — what the training
        forward keeps as `out.raw['matched_distr_kernel_order']`), `targets` [B, T, 3, H, W], `weights` broadcastable [B, T].
        Returns Outputs(dense_img_rec=Outputs(value, weight)) with value = sum over frames and pixels / B (+ `error_mat` [B, T])."""
        from .model import Outputs
        m, hp = self.m, self.m._hp
        est = torch.as_tensor(estimates).to(m.device, torch.float32).contiguous()
        tgt = torch.as_tensor(targets).to(m.device, torch.float32).contiguous()
        B, T = tgt.shape[:2]
        S = hp.img_sz
        w = torch.ones(B, T, device=m.device) * torch.as_tensor(weights, dtype=torch.float32, device=m.device)
        nll_bt = torch.empty(B, T, device=m.device)
        st = torch.cuda.current_stream(m.device).cuda_stream
        rt.check(m.lib.gcpx_dlm_nll(est.data_ptr(), tgt.data_ptr(), w.contiguous().data_ptr(), nll_bt.data_ptr(), B * T, S * S, m._head_pitch,
                                    hp.n_mixtures, st), "dlm_nll")
        res = Outputs(dense_img_rec=Outputs(value=(nll_bt * w).sum() / B, weight=hp.dense_img_rec_weight))
        if log_error_arr:
            res.dense_img_rec.error_mat = nll_bt
        return res

    def decode_seq(self, inputs, enc):
        """DecoderModule.decode_seq(inputs, enc [B, N, nz_enc(,1,1)]): the skips of inputs['I_0'] (or inputs['skips'] from
        `model.encoder`) are broadcast over the N latents of a sequence.  Returns Outputs(images [B, N, 3, H, W])."""
        return self.m._decode_seq(inputs, enc)


class BindingHandle:
    """`model.tree_module.binding` (BalancedBinding, frame_binding.py:37-65) as far as callers outside the forward use it"""

    def __init__(self, model):
        self.m = model

    def get_init_inds(self, outputs):
        """frame_binding.py:62-65: the timesteps of the two virtual root parents, (-1, end_ind + 1), int64 [B, 1]"""
        e = outputs.end_ind if hasattr(outputs, "end_ind") else outputs["end_ind"]
        return torch.zeros_like(e[:, None]) - 1, e[:, None] + 1

    @staticmethod
    def comp_timestep(t_l, t_r):
        """frame_binding.py:52-54 under the torch 1.3 the reference pins: Long / Long truncates toward zero"""
        return torch.div(t_l + t_r, 2, rounding_mode="trunc")


class TreeModuleHandle:
    def __init__(self, model):
        self.binding = BindingHandle(model)


class InverseModelHandle:
    def __init__(self, model):
        self.m = model

    def run_single(self, enc_latent_img0, model_latent_img1):
        """inverse_mdl.py:222-225: action between an encoded frame and a latent the model produced, rows [R, nz_enc] -> [R, n_actions]"""
        return self.m.predictor_rows("inv_mdl", _rows(enc_latent_img0), _rows(model_latent_img1))


class CostModelHandle:
    def __init__(self, model):
        self.m = model

    def cost_pred(self, enc1, enc2):
        return self.m.predictor_rows("cost_mdl", _rows(enc1), _rows(enc2))

    __call__ = cost_pred

    @property
    def input_dim(self):
        return self.m._hp.nz_enc


class DenseRecHandle:
    """TreeDenseRec (tree_dense_rec.py:7-44): evaluation matching of a decoded tree to a dense sequence"""

    def __init__(self, model):
        self.m = model
        self._bindings = {}

    def _binding(self, pruning_scheme):
        """one evaluation binding per pruning scheme (tree_dense_rec.py:8-11 builds the one its hyper-parameter names; callers
        here may ask for several)"""
        b = self._bindings.get(pruning_scheme)
        if b is None:
            from .evaluation import get_eval_binding
            b = self._bindings[pruning_scheme] = get_eval_binding(self.m, pruning_scheme)
        return b

    @property
    def eval_binding(self):
        """the most recently used binding (the reference's attribute name)"""
        return next(reversed(self._bindings.values()), None)

    def get_sample_with_len(self, i_ex, length, outputs, inputs, pruning_scheme, name=None):
        b = self._binding(pruning_scheme)
        return b(outputs, inputs, length, i_ex, name) if pruning_scheme == "basic" else b(outputs, inputs, length, i_ex)

    def get_all_samples_with_len(self, length, outputs, inputs, pruning_scheme, name=None):
        b = self._binding(pruning_scheme)
        if pruning_scheme == "basic":
            return b.get_all_samples(outputs, inputs, length, name)
        return b.get_all_samples(outputs, inputs)


def _rows(x):
    x = torch.as_tensor(x)
    while x.dim() > 2 and x.shape[-1] == 1:            # [R, nz, 1, 1] -> [R, nz]  (remove_spatial)
        x = x[..., 0]
    return x
