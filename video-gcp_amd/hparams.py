"""Hyper-parameters of the gcp_tree hot path.

Values follow the reference's config chain (all paths relative to /root/reference):
  gcp/prediction/hyperparameters.py:4-150            (defaults)
  gcp/prediction/models/auxilliary_models/base_model.py:34-70
  experiments/prediction/base_configs/base_tree.py:11-20
  experiments/prediction/25room/gcp_tree/conf.py:21-43 (25-room gcp_tree overrides)
  gcp/prediction/train.py:80-81                      (hierarchy_levels = ceil(log2(max_seq_len)))

The `blox` building blocks are absent from the reference tree (empty submodule), so everything the
reference delegates to blox is fixed HERE as this build's written spec (see DESIGN.md "Model spec").
"""
import math
from dataclasses import dataclass, field, asdict


@dataclass
class GCPHParams:
    # data
    batch_size: int = 16
    max_seq_len: int = 80
    img_sz: int = 64
    input_nc: int = 3
    state_dim: int = 2
    n_actions: int = 2
    # network size (25room/gcp_tree/conf.py:21-43)
    ngf: int = 16
    nz_enc: int = 128
    nz_mid: int = 128
    nz_vae: int = 256
    nz_mid_lstm: int = 512
    n_lstm_layers: int = 3
    n_processing_layers: int = 3       # hyperparameters.py:21
    init_mlp_layers: int = 3           # hyperparameters.py:27
    init_mlp_mid_sz: int = 32          # hyperparameters.py:28
    conv_inf_enc_kernel_size: int = 3  # hyperparameters.py:22
    conv_inf_enc_layers: int = 1       # hyperparameters.py:23
    nz_attn_key: int = 32              # hyperparameters.py:26
    n_attention_heads: int = 1         # hyperparameters.py:24
    n_attention_layers: int = 1        # hyperparameters.py:25
    # architecture
    use_skips: bool = True
    skips_stride: int = 2
    untied_layers: bool = True
    tree_lstm: str = "split_linear"
    lstm_init: str = "mlp"
    seq_enc: str = "conv"
    context_every_step: bool = True
    matching_type: str = "balanced"    # or "dtw_image" (adaptive binding, base_configs/gcp_adaptive.py:8)
    attentive_inference: bool = False  # base_configs/gcp_adaptive.py:10
    attention_temperature: float = 1.0  # hyperparameters.py:60 (learn_attn_temp=True: a parameter)
    matching_temp: float = 1.0         # hyperparameters.py:94
    learn_matching_temp: bool = True   # hyperparameters.py:132 (adaptive binding only; base_configs/gcp_adaptive.py:9 turns it off)
    learned_pruning_threshold: float = 0.5   # hyperparameters.py:116
    top_bias: float = 1.0              # hyperparameters.py:100
    leaves_bias: float = 0.0           # hyperparameters.py:99
    decoder_distribution: str = "discrete_logistic_mixture"   # or "gaussian"
    n_mixtures: int = 10               # build spec (PixelCNN++ default)
    prior_type: str = "learned"
    regress_length: bool = True
    attach_state_regressor: bool = True
    supervised_decoder: bool = False   # hyperparameters.py:118; consumed at base_gcp.py:252-256 (see run_state_regressor)
    attach_inv_mdl: bool = True
    attach_cost_mdl: bool = True
    run_cost_mdl: bool = True          # hyperparameters.py:63
    train_inv_mdl_full_seq: bool = False   # hyperparameters.py:108
    inv_mdl_temp_dist: int = 1         # inverse_mdl.py:39 (InverseModel default 'temp_dist')
    # flat-predictor variants (gcp_sequential only; experiments/prediction/base_configs/vmpc.py:11-16 sets all four)
    action_conditioned_pred: bool = False    # hyperparameters.py:65: the encoded action of every step feeds the recurrent nets
    non_goal_conditioned: bool = False       # hyperparameters.py:89: goal image (and the sequence's end frame) zeroed, base_gcp.py:163-175
    var_inf: str = "standard"                # hyperparameters.py:80; 'deterministic': no latent (nz_vae = 0); '2layer' is not built
    # build spec for what blox would define
    leaky_slope: float = 0.2
    bn_eps: float = 1e-5
    gn_groups: int = 8
    gn_eps: float = 1e-5
    # loss weights (hyperparameters.py:38-46)
    kl_weight: float = 1.0
    kl_weight_burn_in: float = None    # hyperparameters.py:42, base_gcp.py:121-128: iterations over which the KL weight ramps 0 -> kl_weight
    length_pred_weight: float = 1.0
    dense_img_rec_weight: float = 1.0
    entropy_weight: float = 0.0
    action_rec_weight: float = 1.0     # inverse_mdl.py:56
    free_nats: float = 0.0
    hierarchy_levels: int = field(default=-1)

    def __post_init__(self):
        if self.hierarchy_levels < 0:
            # train.py:80-81
            self.hierarchy_levels = int(math.ceil(math.log2(self.max_seq_len)))
        assert self.img_sz in (32, 64), "decoder kernels are built for 32x32 (reference default, base_model.py:41) and 64x64"
        assert self.nz_mid % self.gn_groups == 0 and self.init_mlp_mid_sz % self.gn_groups == 0
        assert self.decoder_distribution in ("discrete_logistic_mixture", "gaussian")
        assert self.matching_type in ("balanced", "dtw_image")
        assert self.var_inf in ("standard", "deterministic"), "var_inf '2layer' is not built"
        assert (self.var_inf == "deterministic") == (self.nz_vae == 0), "a deterministic predictor has no latent: nz_vae = 0 (vmpc.py:14-15)"
        assert self.nz_attn_key % self.n_attention_heads == 0 and self.nz_enc % self.n_attention_heads == 0

    @property
    def run_state_regressor(self):
        """base_gcp.py:252-256 as written: `regressed_state` is computed INSIDE `if not supervised_decoder:` (the detach of its input and
        the regressor call share the indentation), so with supervised_decoder=True the state regressor — still built, its parameters
        still in the state_dict — produces nothing and the state-regression loss is absent.  Mirrored, not repaired."""
        return self.attach_state_regressor and not self.supervised_decoder

    @property
    def deterministic(self):
        return self.var_inf == "deterministic"

    @property
    def adaptive(self):
        return self.matching_type.startswith("dtw")

    # ---- derived sizes ----
    @property
    def n_nodes(self):
        return 2 ** self.hierarchy_levels - 1

    @property
    def n_conv_layers(self):
        return int(math.log2(self.img_sz))          # blox get_num_layers

    @property
    def lstm_state_dim(self):
        return 2 * self.n_lstm_layers * self.nz_mid_lstm

    @property
    def pred_inp_dim(self):
        d = 2 * self.nz_enc + self.nz_vae           # tree_module.py:40
        if self.context_every_step:
            d += 2 * self.nz_enc                    # tree_module.py:41-42
        return d

    @property
    def head_channels(self):
        if self.decoder_distribution == "gaussian":
            return self.input_nc
        return self.n_mixtures * (1 + 3 * self.input_nc)   # logits + (mean, log_scale, coeff) per colour

    def to_dict(self):
        return asdict(self)


def config(name, **over):
    """Named configurations of BASELINE.json / SURVEY.md §8."""
    base = {
        "c1": dict(batch_size=2, max_seq_len=20, img_sz=32),
        "c2": dict(batch_size=16, max_seq_len=80, img_sz=64),
        "c3": dict(batch_size=16, max_seq_len=80, img_sz=64),      # per-GPU shard of 128
        "c4": dict(batch_size=64, max_seq_len=80, img_sz=64),      # per-GPU shard of 512 candidates
        "c5": dict(batch_size=8, max_seq_len=200, img_sz=64,       # per-GPU shard of 64; adaptive binding + attentive
                   matching_type="dtw_image", attentive_inference=True,        # inference (base_configs/gcp_adaptive.py:6-11)
                   learn_matching_temp=False),
        "c5s": dict(batch_size=2, max_seq_len=12, img_sz=32,       # small adaptive case for parity tests (L=4, N=15)
                    matching_type="dtw_image", attentive_inference=True, learn_matching_temp=False),
        # the visual-MPC style flat predictor of base_configs/vmpc.py at the parity-test size (gcp_sequential only)
        "vmpc_s": dict(batch_size=2, max_seq_len=8, img_sz=32, action_conditioned_pred=True, non_goal_conditioned=True,
                       nz_vae=0, var_inf="deterministic"),
        "tiny": dict(batch_size=2, max_seq_len=6, img_sz=32, nz_mid_lstm=64, n_lstm_layers=2,
                     nz_vae=32, nz_enc=32, nz_mid=32),
    }[name]
    base.update(over)
    return GCPHParams(**base)
