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

    def decode_seq(self, inputs, enc):
        """DecoderModule.decode_seq(inputs, enc [B, N, nz_enc(,1,1)]): the skips of inputs['I_0'] (or inputs['skips'] from
        `model.encoder`) are broadcast over the N latents of a sequence.  Returns Outputs(images [B, N, 3, H, W])."""
        return self.m._decode_seq(inputs, enc)


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
        self.eval_binding = None

    def _binding(self, pruning_scheme):
        if self.eval_binding is None:
            from .evaluation import get_eval_binding
            self.eval_binding = get_eval_binding(self.m, pruning_scheme)
        return self.eval_binding

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
