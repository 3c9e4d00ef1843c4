"""Packed weights of the backward pass: transposed (and flipped) MFMA-order packs for the data gradients, their split-f16 twins, and the
live re-split of the wide tree levels' GEMM weights (BackwardWeightsMixin, mixed into training.GCPTrainStep)."""
import ctypes as C
import os
import re

import torch

from . import packing as pk
from . import runtime as rt
from .plan_ops import _Plan, _addr, N_LANES
from .params import decoder_layers


def _c16(n):
    return (n + 15) // 16 * 16


class BackwardWeightsMixin:

    # ------------------------------------------------------------------------------------------------
    # transposed weight packs (data-gradient GEMMs / convs)
    # ------------------------------------------------------------------------------------------------
    def _pack_predictor_T(self, sd, prefix, splits):
        """splits: list of (col0, width) groups of the input layer whose gradients go to different places."""
        T = {}
        w_out = sd[f"{prefix}.head.linear.weight"]
        od = w_out.shape[0]
        w_out = pk._pad_rows(w_out, _c16(od))
        T["wT_out"] = pk.pack_gemm(w_out.t().contiguous())                 # [N = mid][K = out_pad]
        l = 0
        while f"{prefix}.pyramid-{l}.linear.weight" in sd:
            T[f"wT_mid{l}"] = pk.pack_gemm(sd[f"{prefix}.pyramid-{l}.linear.weight"].t().contiguous())
            l += 1
        w_in = sd[f"{prefix}.input.linear.weight"]
        for i, (c0, w) in enumerate(splits):
            T[f"wT_in{i}"] = pk.pack_gemm(w_in[:, c0:c0 + w].t().contiguous())   # [N = w][K = mid]
        return T

    def _pack_backward_split(self):
        """Split-f16 pieces of the transposed, flipped weights of the data-gradient convs that run the wave-autonomous kernel
        (conv3x3_wave_split_kernel): the output head's (112 kernel slots -> 16) and the 16-channel decoder blocks' (16 -> 32).  Like
        the forward's split weights they are index gathers of the flat parameter vector (model.pk_split), re-split by
        gcpx_split_pack behind every optimizer step."""
        m, hp = self.m, self.m._hp
        if not m.split_f16:
            return

        def ids_of(key):
            off, shp = m._poff[key]
            n = 1
            for d in shp:
                n *= d
            return (torch.arange(n, dtype=torch.float64) + (off + 1)).view(shp)

        todo = {}
        if hp.decoder_distribution == "discrete_logistic_mixture":
            hw = ids_of("decoder.gen_head.conv.weight")                      # [100, 16, 3, 3]; inputs of the dgrad = kernel slots
            perm = torch.as_tensor(pk.dlm_channel_perm(hp.n_mixtures))
            wk = torch.zeros((len(perm),) + tuple(hw.shape[1:]), dtype=hw.dtype)
            wk[perm >= 0] = hw[perm[perm >= 0]]
            todo["bw.dec.head"] = wk.flip(2, 3).transpose(0, 1).contiguous()
        for name, c_prev, c_skip, skip_idx, cout in decoder_layers(hp):
            if cout == 16 and c_prev + c_skip == 32:
                todo[f"bw.dec.{name}"] = ids_of(f"decoder.net.{name}.conv.weight").flip(2, 3).transpose(0, 1).contiguous()
            elif self.split_dgrad_wide and cout % 16 == 0 and (c_prev + c_skip) % 32 == 0:
                # wider blocks: 32 of the block's input channels per launch on the same kernel (conv3x3_wave_split_kernel<2>: the exact
                # f32 tiles they ran on take 3x the MFMA time), where the frame is >= 16 wide (_decoder_backward)
                wT = ids_of(f"decoder.net.{name}.conv.weight").flip(2, 3).transpose(0, 1).contiguous()          # [cin, cout, 3, 3]
                for h in range((c_prev + c_skip) // 32):
                    todo[f"bw.dec.{name}.q{h}"] = wT[32 * h:32 * (h + 1)].contiguous()
        for name, wT in todo.items():
            idx = (pk.conv3x3_split_gather(wT).reshape(-1) - 1).to(torch.int32).to(m.device)
            m.pk_split[name] = dict(idx=idx, out=torch.zeros(2 * idx.numel(), dtype=torch.int16, device=m.device),
                                    log2=torch.zeros(1, dtype=torch.int32, device=m.device))
        m.repack_split()

    def _live_gemm_split(self):
        """The GEMM weights of the tree levels that run with >= 512 rows at the configured batch size, kept in split-f16 form ALSO in
        training (the inference model splits them once at weight load, model._pack_gemm_split): their forward merge / output GEMMs and every
        data-gradient GEMM of those levels then run on the split-f16 kernels (3x the f32 MFMA rate) instead of the exact f32 tiles.  Each
        pack is an index gather of the flat parameter vector — the arena's index map pushed through packing.unpack_gemm / gemm_split_gather —
        re-split by gcpx_split_pack_group2 with the slice of the optimizer step it belongs to (model.repack(bucket=): under the encoder
        backward, like the slice itself)."""
        m, hp = self.m, self.m._hp
        m._gsplit, m._gsplit_tabs, m._gsplit_live = {}, {}, False
        if not (m.split_f16 and hp.tree_lstm and os.environ.get("GCPX_NO_LIVE_GEMM_SPLIT") is None):
            return
        min_rows = int(os.environ.get("GCPX_GEMM_SPLIT_MIN_ROWS", "512"))
        L = hp.hierarchy_levels
        levels = [l for l in range(L) if hp.batch_size * 2 ** l >= min_rows and f"tree{l}" in m.pk and f"tree{l}" in self.bk]
        if not hp.untied_layers:
            levels = [0] if levels else []
        base = m._arena.data_ptr()
        descs = {}
        keep = []
        for l in levels:
            bucket = f"tree{l}" if f"tree{l}" in [n_ for n_, _, _ in m._arena_ranges] else m._arena_ranges[-1][0]

            fwd = lambda k: k in ("proj.w", "out.w", "embed.w") or re.fullmatch(r"lstm\d+\.w", k)
            bwd = lambda k: k in ("proj.wT", "out.wT", "embed.wT", "lstm.whT") or re.fullmatch(r"lstm\d+\.wxT", k)
            leaves = [(k, v) for k, v in m.pk[f"tree{l}"].items() if torch.is_tensor(v) and fwd(k)]
            leaves += [(k, v) for k, v in self.bk[f"tree{l}"].items() if torch.is_tensor(v) and bwd(k)]
            for k, leaf in leaves:
                stack = leaf if leaf.dim() == 5 else leaf[None]
                KG, NT = stack.shape[1], stack.shape[2]
                N, K = NT * 16, KG * 16
                if K % 64 or N % 64:
                    continue
                off = (leaf.data_ptr() - base) // 4
                idx0 = m._arena_idx0[off:off + leaf.numel()].view(stack.shape).to(torch.int64)
                n_el = N * K
                ws = torch.zeros(stack.shape[0], 2 * n_el, dtype=torch.int16, device=m.device)
                es = torch.zeros(stack.shape[0], dtype=torch.int32, device=m.device)
                for b in range(stack.shape[0]):
                    ids = pk.gemm_split_gather(pk.unpack_gemm(idx0[b] + 1, N)).reshape(-1) - 1          # (-1: a zero-padded slot)
                    ids = ids.to(torch.int32).contiguous()
                    e = rt.SplitPackDesc()
                    e.src, e.idx, e.out, e.log2_out, e.n = m.theta.data_ptr(), ids.data_ptr(), ws[b].data_ptr(), es[b:b + 1].data_ptr(), n_el
                    descs.setdefault(bucket, []).append(e)
                    keep.append(ids)
                m._gsplit[leaf.data_ptr()] = (ws, es)
        self._gsplit_keep = keep
        for bucket, ds in descs.items():
            arr = (rt.SplitPackDesc * len(ds))(*ds)
            dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(m.device)
            m._gsplit_tabs[bucket] = (dev, len(ds), torch.zeros(len(ds), dtype=torch.int32, device=m.device))
        m._gsplit_live = bool(descs)
        m._repack_gsplit(torch.cuda.current_stream(m.device).cuda_stream, None)

    def _pack_backward(self, sd):
        m, hp = self.m, self.m._hp
        nz, nv, H = hp.nz_enc, hp.nz_vae, hp.nz_mid_lstm
        X = {}
        layers, ctop = m._enc_layers, m._c_top
        for name, cin, cout, norm in layers[1:]:
            w = sd[f"encoder.net.{name}.conv.weight"]                       # [co, ci, 4, 4] -> [n = (tap, ci)][k = co]
            X[f"enc.{name}.wT"] = pk.pack_gemm(w.permute(2, 3, 1, 0).reshape(16 * cin, cout))
        wh = sd["encoder.net.head.weight"]                                   # [nz, c, 4, 4] -> [n = (tap, c)][k = nz]
        X["enc.head.wT"] = pk.pack_gemm(wh.permute(2, 3, 1, 0).reshape(16 * ctop, nz))
        wt = sd["decoder.net.input.conv.weight"]                             # [nz, co, 4, 4] -> [n = nz][k = (tap, co)]
        X["dec.input.wT"] = pk.pack_gemm(wt.permute(0, 2, 3, 1).reshape(nz, 16 * ctop))
        for name, c_prev, c_skip, skip_idx, cout in decoder_layers(hp):
            w = sd[f"decoder.net.{name}.conv.weight"]                       # dgrad = 3x3 conv with w^T flipped
            wT = w.flip(2, 3).transpose(0, 1).contiguous()                  # [cin, cout, 3, 3]
            cin = wT.shape[0]
            for h in range((cin + 63) // 64):
                X[f"dec.{name}.wT{h}"] = pk.pack_conv3x3(wT[64 * h:64 * (h + 1)], 16)
            if self.split_dgrad_wide and m.split_f16 and cin % 32 == 0 and not (cout == 16 and cin == 32):
                # (the f32 twin of the 32-channel launches of _decoder_backward: what the launch falls back to if its split form does not fit)
                for h in range(cin // 32):
                    X[f"dec.{name}.wTq{h}"] = pk.pack_conv3x3(wT[32 * h:32 * (h + 1)], 16)
        hw = sd["decoder.gen_head.conv.weight"]                              # [100, 16, 3, 3]; inputs of the dgrad = kernel slots
        perm = torch.as_tensor(pk.dlm_channel_perm(hp.n_mixtures), device=hw.device)
        wk = torch.zeros((len(perm),) + tuple(hw.shape[1:]), dtype=hw.dtype, device=hw.device)
        wk[perm >= 0] = hw[perm[perm >= 0]]
        X["dec.head.wT"] = pk.pack_conv3x3(wk.flip(2, 3).transpose(0, 1).contiguous(), 16)
        for nm in (["input"] + [f"pyramid-{i}" for i in range(hp.conv_inf_enc_layers)] + ["head"]) if hp.seq_enc == "conv" else []:
            w = sd[f"inf_encoder.net.{nm}.conv.weight"]                     # [co, ci, 3] -> [n = ci][k = (tap, co)]
            X[f"seq.{nm}.wT"] = pk.pack_gemm(w.permute(1, 2, 0).reshape(w.shape[1], -1))
        if hp.regress_length:
            X["length_pred"] = self._pack_predictor_T(sd, "length_pred.p", [(0, 2 * nz)])
        if hp.attach_state_regressor:
            X["state_regressor"] = self._pack_predictor_T(sd, "state_regressor", [])
        if hp.attach_inv_mdl:
            X["inv_mdl"] = self._pack_predictor_T(sd, "inv_mdl.action_pred", [])
        if hp.attach_cost_mdl:
            X["cost_mdl"] = self._pack_predictor_T(sd, "cost_mdl.cost_pred", [])
        if hp.adaptive:
            X["distance"] = self._pack_predictor_T(sd, "tree_module.tree_modules.0.binding.distance_predictor", [(0, nz), (nz, nz)])
        else:
            X["existence"] = self._pack_predictor_T(sd, "tree_module.tree_modules.0.binding.existence_predictor", [(0, nz)])
        if hp.attentive_inference:
            for nm in (["input"] + [f"pyramid-{i}" for i in range(hp.conv_inf_enc_layers)] + ["head"]) if hp.seq_enc == "conv" else []:
                w = sd[f"inf_key_encoder.0.net.{nm}.conv.weight"]
                X[f"kseq.{nm}.wT"] = pk.pack_gemm(w.permute(1, 2, 0).reshape(w.shape[1], -1))
            X["kseq.key.wT"] = pk.pack_gemm(sd["inf_key_encoder.1.linear.weight"].t().contiguous())                # [nz][dk]
            n_mod = hp.hierarchy_levels if hp.untied_layers else 1
            att = lambda l, nm: sd[f"tree_module.tree_modules.{l}.inference.attention.attention_layers.0.{nm}.weight"]
            # d keys = [dK'_0 | dK'_1 | ...] @ [Wk_0; Wk_1; ...]: one GEMM over the level blocks laid side by side
            X["attn.k_proj.wT"] = pk.pack_gemm(torch.cat([att(l, "k_proj") for l in range(n_mod)], 0).t().contiguous())   # [dk][n_mod*dk]
            X["attn.v_proj.wT"] = pk.pack_gemm(torch.cat([att(l, "v_proj") for l in range(n_mod)], 0).t().contiguous())   # [nz][n_mod*nz]
        for l in range(hp.hierarchy_levels if hp.untied_layers else 1):
            p = f"tree_module.tree_modules.{l}"
            T = {}
            T["prior"] = self._pack_predictor_T(sd, f"{p}.prior", [(0, 2 * nz)])
            T["q"] = self._pack_predictor_T(sd, f"{p}.inference.q", [(0, 2 * nz), (2 * nz, nz)])
            if not hp.tree_lstm:                                   # non-LSTM subgoal predictor (tree_module.py:109-110)
                T["sg"] = self._pack_predictor_T(sd, f"{p}.subgoal_pred.net", [(0, hp.pred_inp_dim)])
                X[f"tree{l}"] = T
                continue
            T["embed.wT"] = pk.pack_gemm(sd[f"{p}.subgoal_pred.embed.weight"].t().contiguous())
            for i in range(hp.n_lstm_layers):
                T[f"lstm{i}.wxT"] = pk.pack_gemm(sd[f"{p}.subgoal_pred.lstm.{i}.weight_ih"].t().contiguous())   # [H][4H]
            # (stacked: the layers' d h_prev GEMMs of a level are ONE batched launch behind the level's d x chain)
            T["lstm.whT"] = torch.stack([pk.pack_gemm(sd[f"{p}.subgoal_pred.lstm.{i}.weight_hh"].t().contiguous())
                                         for i in range(hp.n_lstm_layers)]).contiguous()
            T["out.wT"] = pk.pack_gemm(sd[f"{p}.subgoal_pred.out.weight"].t().contiguous())                     # [H][nz]
            if hp.tree_lstm == "split_linear":
                T["proj.wT"] = torch.stack([pk.pack_gemm(sd[f"{p}.subgoal_pred.projections.{j}.weight"].t().contiguous())
                                            for j in range(2 * hp.n_lstm_layers)]).contiguous()                 # [2H][H] each
            elif hp.tree_lstm == "linear":
                T["proj.wT"] = pk.pack_gemm(sd[f"{p}.subgoal_pred.projection.weight"].t().contiguous())         # [n = 2 SD][k = SD]
            if l == 0 and hp.lstm_init == "mlp":
                T["init"] = self._pack_predictor_T(sd, f"{p}.lstm_initializer.net", [(0, 2 * nz + nv)])
            if hp.attentive_inference:
                a = f"{p}.inference.attention"
                T["attn.query"] = self._pack_predictor_T(sd, f"{a}.query_net", [(0, 2 * nz)])
                T["attn.q_proj.wT"] = pk.pack_gemm(sd[f"{a}.attention_layers.0.q_proj.weight"].t().contiguous())
                T["attn.out_proj.wT"] = pk.pack_gemm(sd[f"{a}.attention_layers.0.out_proj.weight"].t().contiguous())
                T["attn.out.wT"] = pk.pack_gemm(sd[f"{a}.out.weight"].t().contiguous())
            X[f"tree{l}"] = T
        return X
