"""RoomFormerV2 base model of CAPE on MI355X kernels: backbone -> input_proj (1x1 / 3x3-s2 conv as
implicit GEMM + GroupNorm written straight into the flattened token buffer) -> deformable
transformer; teacher-forced `forward` and KV-cached autoregressive `forward_inference` whose token
bookkeeping runs on the device.  Interface and parameter names follow the reference
(`models/roomformer_v2.py:149-693, :956-1049`)."""
import copy
import math
import os
import warnings

import torch
from torch import nn

from ..hip import functional as HF
from ..hip import ops
from ..util.misc import NestedTensor, cached_zero_mask, nested_tensor_from_tensor_list
from .backbone import Conv2dCL, build_backbone
from .deformable_transformer_v2 import (DecodeWeights, alloc_decode_workspace, build_deforamble_transformer, decode_step_fused)
from .kv_cache import KVCache, VCache


def _get_clones(module, N):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(N)])


class MLP(nn.Module):
    """Very simple multi-layer perceptron (roomformer_v2.py:956-968); applied through HIP GEMMs."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(nn.Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))

    def forward(self, x):
        for i, layer in enumerate(self.layers):
            x = HF.linear(x, layer.weight, layer.bias, relu=(i < self.num_layers - 1))
        return x


class RoomFormerV2(nn.Module):
    def __init__(self, backbone, transformer, num_classes, num_queries, num_polys, num_feature_levels, aux_loss=True,
                 with_poly_refine=False, masked_attn=False, semantic_classes=-1, seq_len=1024, tokenizer=None,
                 use_anchor=False, patch_size=1, freeze_anchor=False, inject_cls_embed=False, cape_mode=False):
        super().__init__()
        assert num_queries % num_polys == 0
        if cape_mode:
            raise ValueError("cape_mode=True (RoomFormerV2-internal SupportPoseEncoder) is not on the CAPE path; "
                             "models.build_model always passes cape_mode=False")
        if not with_poly_refine or use_anchor or inject_cls_embed or num_feature_levels != 4:
            raise ValueError("CAPE path: with_poly_refine=True, use_anchor=False, inject_cls_embed=False, 4 feature levels")
        self.num_queries, self.num_polys = num_queries, num_polys
        self.transformer = transformer
        hidden_dim = transformer.d_model
        self.num_classes = num_classes
        self.cape_mode = cape_mode
        self.class_embed = nn.Linear(hidden_dim, num_classes)
        self.coords_embed = MLP(hidden_dim, hidden_dim, 2, 3)
        self.num_feature_levels = num_feature_levels
        self.tokenizer = tokenizer
        self.seq_len = seq_len
        self._decode_states = {}       # batch geometry -> static decode buffers + captured step graphs (forward_inference)
        self.patch_size = patch_size
        self.inject_cls_embed = inject_cls_embed
        num_backbone_outs = len(backbone.strides)
        proj = []
        in_channels = None
        for i in range(num_backbone_outs):
            in_channels = backbone.num_channels[i]
            proj.append(nn.Sequential(Conv2dCL(in_channels, hidden_dim, patch_size, stride=patch_size, padding=0, bias=True),
                                      nn.GroupNorm(32, hidden_dim)))
        for _ in range(num_feature_levels - num_backbone_outs):
            if patch_size == 1:
                proj.append(nn.Sequential(Conv2dCL(in_channels, hidden_dim, 3, stride=2, padding=1, bias=True),
                                          nn.GroupNorm(32, hidden_dim)))
            else:
                proj.append(nn.Sequential(Conv2dCL(in_channels, hidden_dim, 2 * patch_size, stride=2 * patch_size, padding=0,
                                                   bias=True), nn.GroupNorm(32, hidden_dim)))
            in_channels = hidden_dim
        self.input_proj = nn.ModuleList(proj)
        self.backbone = backbone
        self.aux_loss = aux_loss
        self.with_poly_refine = with_poly_refine

        prior_prob = 0.01
        bias_value = -math.log((1 - prior_prob) / prior_prob)
        self.class_embed.bias.data = torch.ones(num_classes) * bias_value
        nn.init.constant_(self.coords_embed.layers[-1].weight.data, 0)
        nn.init.constant_(self.coords_embed.layers[-1].bias.data, 0)
        for p in self.input_proj:
            w = torch.empty_like(p[0].weight, memory_format=torch.contiguous_format)
            nn.init.xavier_uniform_(w, gain=1)
            with torch.no_grad():
                p[0].weight.copy_(w)
            nn.init.constant_(p[0].bias, 0)
        num_pred = transformer.decoder.num_layers
        self.class_embed = _get_clones(self.class_embed, num_pred)
        self.coords_embed = _get_clones(self.coords_embed, num_pred)
        nn.init.constant_(self.coords_embed[0].layers[-1].bias.data[2:], -2.0)
        self.query_embed = nn.Embedding(seq_len, 2)
        self.query_embed.weight.requires_grad = not freeze_anchor
        self.transformer.decoder.coords_embed = self.coords_embed
        self.transformer.decoder.class_embed = self.class_embed
        self.room_class_embed = None
        if semantic_classes > 0:
            self.room_class_embed = nn.Linear(hidden_dim, semantic_classes)
        self.register_buffer("attention_mask", self._create_causal_attention_mask(seq_len))

    @staticmethod
    def _create_causal_attention_mask(seq_len):
        mask = torch.triu(torch.ones(seq_len, seq_len), diagonal=1)
        return mask.masked_fill(mask == 1, float("-inf")).masked_fill(mask == 0, 0.0)

    # ---- image side shared by forward / forward_inference ----------------------------------------
    def _encode_images(self, samples):
        has_padding = True
        if not isinstance(samples, NestedTensor):
            has_padding = not (isinstance(samples, torch.Tensor) and samples.ndim == 4)
            samples = nested_tensor_from_tensor_list(samples)
        elif samples.mask is not None:
            has_padding = True if HF.capturing() else bool(samples.mask.any())     # no host sync inside a graph capture
        if not has_padding:
            samples.no_padding = True            # equally sized images: every resampled mask below is the cached all-False one
        features = self.backbone(samples)
        srcs, masks = [], []
        last = None

        def level_mask(mask, src):
            if not has_padding:
                return cached_zero_mask(src.shape[0], src.shape[1], src.shape[2], src.device)
            return _nearest_mask(mask, src.shape[1], src.shape[2])

        for l, feat in enumerate(features):
            x, mask = feat.decompose()
            last = x
            if l == len(features) - 1 and self.num_feature_levels > len(features):
                x, last = HF.fanout(x, 2)        # C5 also feeds the extra stride-2 level
            conv = self.input_proj[l][0]
            src = HF.conv_bn_act(x, conv.weight, None, conv.bias, conv.stride, conv.padding, relu=False)
            if self.patch_size != 1:
                mask = level_mask(mask, src)
            srcs.append(src); masks.append(mask)
        for l in range(len(features), self.num_feature_levels):
            conv = self.input_proj[l][0]
            inp = last if l == len(features) else srcs[-1]
            src = HF.conv_bn_act(inp, conv.weight, None, conv.bias, conv.stride, conv.padding, relu=False)
            masks.append(level_mask(samples.mask, src))
            srcs.append(src)
        gammas = [p[1].weight for p in self.input_proj]
        betas = [p[1].bias for p in self.input_proj]
        return self.transformer.encode(srcs, (gammas, betas), masks, has_padding)

    def forward(self, samples, seq_kwargs=None, support_graphs=None, support_mask=None):
        """Teacher-forced pass.  Returns {'pred_logits', 'pred_coords', 'pred_room_logits', 'aux_outputs'}."""
        enc = self._encode_images(samples)
        hs, init_reference, inter_references, inter_classes = self.transformer(enc, self.query_embed.weight, seq_kwargs)
        out = {"pred_logits": inter_classes[-1], "pred_coords": inter_references[-1]}
        if self.room_class_embed is not None:
            out["pred_room_logits"] = HF.linear(hs[-1], self.room_class_embed.weight, self.room_class_embed.bias)
        if self.aux_loss:
            out["aux_outputs"] = [{"pred_logits": a, "pred_coords": b}
                                  for a, b in zip(inter_classes[:-1], inter_references[:-1])]
        # kept for the criterion: stacked per-layer outputs avoid re-stacking in the loss
        out["_stack_logits"], out["_stack_coords"] = inter_classes, inter_references
        return out

    def _decode_plan(self, dec, dw, caches, emb, vr, geo, N):
        """Descriptor of the whole-step decode kernel: the pointers of every decoder weight, cache and table (ops.DecodeStepPlan)."""
        layers = []
        for l, (layer, w, c) in enumerate(zip(dec.layers, dw, caches)):
            sa, ca, m = layer.self_attn, layer.support_attn, layer.cross_attn
            mlp = dec.coords_embed[l].layers
            layers.append({
                "w_qkv": w["w_qkv"], "b_qkv": sa.in_proj_bias, "w_qin": sa.in_proj_weight[:256], "k_cache": c["k"], "v_cache": c["v"],
                "w_o": sa.out_proj.weight, "b_o": sa.out_proj.bias, "ln2_g": layer.norm2.weight, "ln2_b": layer.norm2.bias,
                "w_sq": ca.in_proj_weight[:256], "b_sq": ca.in_proj_bias[:256], "sup_k": c["sup_k"], "sup_v": c["sup_v"],
                "sup_mask": c["sup_kpm"], "w_so": ca.out_proj.weight, "b_so": ca.out_proj.bias,
                "lns_g": layer.norm_support.weight, "lns_b": layer.norm_support.bias,
                "w_off": w["w_off"], "b_off": w["b_off"], "value": c["value"], "w_mo": m.output_proj.weight, "b_mo": m.output_proj.bias,
                "ln1_g": layer.norm1.weight, "ln1_b": layer.norm1.bias, "w1": layer.linear1.weight, "b1": layer.linear1.bias,
                "w2": layer.linear2.weight, "b2": layer.linear2.bias, "ln3_g": layer.norm3.weight, "ln3_b": layer.norm3.bias,
                "m1w": mlp[0].weight, "m1b": mlp[0].bias, "m2w": mlp[1].weight, "m2b": mlp[1].bias, "m3w": mlp[2].weight, "m3b": mlp[2].bias})
        ce = dec.class_embed[len(dec.layers) - 1]
        return ops.DecodeStepPlan(N, self.seq_len, geo, dec.layers[0].cross_attn.n_points, dec.layers[0].linear1.weight.shape[0], emb, vr,
                                  ops.dim_t(emb.device), (ce.weight, ce.bias),
                                  (dec.pos_trans.weight, dec.pos_trans.bias, dec.pos_trans_norm.weight, dec.pos_trans_norm.bias), layers)

    # ---- KV-cached autoregressive inference -------------------------------------------------------
    @torch.no_grad()
    def forward_inference(self, samples, use_cache=True, support_graphs=None, support_mask=None, sync_every=8,
                          teacher_stream=None, graph=None, timing=False):
        """Generates until every sample has emitted <eos> (after >= 6 steps) or `tokenizer.seq_len` steps.
        Token bookkeeping (roomformer_v2.py:521-598) runs on the device; the host only polls the
        `unfinished` flags every `sync_every` steps, then trims to the step at which the reference's loop
        would have stopped (identical outputs: steps after the stop only feed <pad> tokens).
        `teacher_stream` (dict of (N,T) token/delta tensors) replaces the model's own feedback for parity tests.

        graph (default: env CAPE_DECODE_GRAPH, on): a decode step is ~170 launches of 32-row kernels, i.e. host-bound
        (2.7 ms per step against ~0.7 ms of GPU work), so the steps are captured as hipGraphs -- one per step index,
        because the cache row written, the attention length and the output slot are baked into the launch arguments -- over
        a set of static state buffers per batch geometry.  The first call for a geometry runs eagerly, the second
        captures while it decodes, later calls replay."""
        if not use_cache:
            raise ValueError("the MI355X path always decodes with caches (use_cache=False is a debugging mode of the reference)")
        if graph is None:
            graph = os.environ.get("CAPE_DECODE_GRAPH", "1") == "1"
        graph = graph and teacher_stream is None
        # fused step (csrc/decode_step.hip): ~75 launches instead of ~190; CAPE_DECODE_FUSED=0 keeps the per-op step for A/B
        fused = os.environ.get("CAPE_DECODE_FUSED", "1") == "1"
        enc = self._encode_images(samples)
        dec = self.transformer.decoder
        geo, vr, memory = enc["geo"], enc["valid_ratios"], enc["memory"]
        N, dev = memory.shape[0], memory.device
        tok = self.tokenizer
        max_len = tok.seq_len if teacher_stream is None else min(tok.seq_len, teacher_stream["seq11"].shape[1])
        min_len = 6
        support = getattr(dec, "support_features", None)
        smask = getattr(dec, "support_mask", None)
        P = support.shape[1] if support is not None else 0

        # ---- state buffers (static per geometry when graphs are used) ----
        # whole-step kernel (csrc/decode_fused.hip, one block per image, ONE launch per step): needs the CAPE layer shape
        # (support attention in every layer, 4 levels x 4 points, dim_feedforward 1024); CAPE_DECODE_MEGA=0 keeps the
        # launch-per-stage step for A/B
        m0 = dec.layers[0].cross_attn
        mega = (fused and os.environ.get("CAPE_DECODE_MEGA", "1") == "1" and P > 0 and geo.L * m0.n_points == 16 and
                dec.layers[0].linear1.weight.shape[0] == 1024 and self.seq_len <= 1024 and P <= 1024 and geo.S < 65535 and
                len(dec.layers) <= 8 and self.num_classes <= 8)
        fused = fused and (N <= 64 or mega)
        key = (N, tuple(geo.shapes), P, smask is not None, max_len, str(dev), ops.get_gemm_precision(), fused, mega)
        st = self._decode_states.get(key) if graph else None
        fresh = st is None
        if fresh:
            st = {"calls": 0, "graphs": {}, "pool": None,
                  "vr": torch.empty_like(vr), "ref_all": torch.empty(self.query_embed.weight.shape[0], 2, device=dev),
                  "toks": torch.empty(4, N, dtype=torch.int64, device=dev), "deltas": torch.empty(4, N, device=dev),
                  "unfinished": torch.empty(N, dtype=torch.int32, device=dev), "step_t": torch.zeros(1, dtype=torch.int32, device=dev),
                  "out_logits": torch.zeros(N, max_len, self.num_classes, device=dev), "out_coords": torch.zeros(N, max_len, 2, device=dev),
                  "out_hs": torch.zeros(N, max_len, 256, device=dev), "alive_after": torch.zeros(max_len, dtype=torch.int32, device=dev),
                  # the reference's cache modules (kv_cache.py): K / V slabs (N, seq_len, 256) written in place at row `step`
                  # (here: post-projection rows), VCache = the per-layer MSDA value projection of the image memory
                  "kv": [KVCache(N, self.seq_len, 256, torch.float32).to(dev) for _ in dec.layers],
                  "vc": [VCache(N, geo.S, self.transformer.nhead, 256 // self.transformer.nhead, torch.float32).to(dev) for _ in dec.layers]}
            st["caches"] = [{"k": kv.k_cache, "v": kv.v_cache, "value": vc.v_cache.view(N, geo.S, 256),
                             "sup_k": torch.empty(N, P, 256, device=dev) if P else None,
                             "sup_v": torch.empty(N, P, 256, device=dev) if P else None,
                             "sup_kpm": torch.empty(N, P, dtype=torch.uint8, device=dev) if (P and smask is not None) else None}
                            for kv, vc in zip(st["kv"], st["vc"])]
            if fused:
                st["ws"] = alloc_decode_workspace(N, len(dec.layers), geo.L, dev)
                st["alive_i32"] = torch.zeros(max_len, dtype=torch.int32, device=dev)
                # layer-0 tables (static storage: captured step graphs hold pointers into them)
                st["qpos0"] = torch.empty(self.query_embed.weight.shape[0], 256, device=dev)
                st["ref0"] = torch.empty(max_len, N, 2, device=dev)
                st["refin0"] = torch.empty(max_len, N, geo.L, 2, device=dev)
            if graph:
                if len(self._decode_states) >= 4:
                    self._decode_states.pop(next(iter(self._decode_states)))
                self._decode_states[key] = st
        # the captured steps hold raw weight pointers: if the parameters were re-homed since (model.to(), arena creation,
        # load into new storage) the graphs are dropped and re-captured
        sentinel = tuple(p_.data_ptr() for p_ in list(dec.parameters())[:4]) + (self.query_embed.weight.data_ptr(),)
        if fused:
            # the folded inference weights are tensors of their own: captured steps (and the whole-step descriptor) point at them,
            # so their identity (rebuilt after every optimizer step / weight load) is part of what invalidates the graphs
            if getattr(self, "_decode_weights", None) is None:
                self._decode_weights = DecodeWeights(dec)
            self._decode_weights.get()
            sentinel = sentinel + (self._decode_weights.key,)
        if st.get("weights_at") != sentinel:
            st["graphs"], st["pool"], st["weights_at"] = {}, None, sentinel
            st["calls"] = 0
        st["calls"] += 1
        caches = st["caches"]
        for layer, kv, vc in zip(dec.layers, st["kv"], st["vc"]):      # the modules the reference's _setup_caches installs
            layer.kv_cache, layer.cross_attn.cache = kv, vc
        for layer, c in zip(dec.layers, caches):
            c["value"].copy_(layer.cross_attn.project_value(memory, enc["pad_rows"]))
            if P:
                ca = layer.support_attn
                s2 = support.contiguous().view(N * P, 256)
                ops.gemm(s2, ca.in_proj_weight[256:], c["sup_k"], N * P, 256, 256, bias=ca.in_proj_bias[256:])
                ops.gemm(s2, ca.in_proj_weight[512:], c["sup_v"], N * P, 256, 256, bias=ca.in_proj_bias[512:])
                if c["sup_kpm"] is not None:
                    c["sup_kpm"].copy_(smask.to(torch.uint8))
        st["vr"].copy_(vr)
        st["ref_all"].copy_(ops.sigmoid_fwd(self.query_embed.weight.detach().contiguous()))            # (seq_len, 2)
        toks, deltas, unfinished, step_t = st["toks"], st["deltas"], st["unfinished"], st["step_t"]
        toks.fill_(tok.bos)
        deltas.copy_(torch.tensor([0.0, 1.0, 0.0, 1.0], device=dev).view(4, 1).expand(4, N))
        unfinished.fill_(1)
        out_logits, out_coords, out_hs, alive_after = st["out_logits"], st["out_coords"], st["out_hs"], st["alive_after"]
        vr_s, ref_all = st["vr"], st["ref_all"]

        def step_body(i, toks_i, deltas_i):
            ref_i = ref_all[i].view(1, 1, 2).expand(N, 1, 2).contiguous()
            hs, ref, cls = dec.decode_step(toks_i, deltas_i, ref_i, geo, vr_s, i, caches)
            out_logits[:, i] = cls; out_coords[:, i] = ref.view(N, 2); out_hs[:, i] = hs.view(N, 256)
            step_t.fill_(i)
            ops.decode_next_tokens(cls, ref.view(N, 2), unfinished, toks, deltas, step_t, N, tok.num_bins, min_len,
                                   tok.eos, tok.sep, tok.pad)
            alive_after[i] = unfinished.sum()

        if fused:
            # per-call tables of layer 0 (its reference points are the learned anchors, the same for every image):
            # query position embedding per step, level-scaled points and the (N, 2) reference rows the tail kernel refines
            if getattr(self, "_decode_weights", None) is None:
                self._decode_weights = DecodeWeights(dec)
            dw = self._decode_weights.get()
            T0 = ref_all.shape[0]
            qs0 = ops.query_sine_fwd(ref_all)
            qp0 = torch.empty(T0, 256, device=dev)
            ops.gemm(qs0, dec.pos_trans.weight, qp0, T0, 256, 256, bias=dec.pos_trans.bias)
            qp0, _, _, _ = ops.add_layernorm_fwd(qp0, None, dec.pos_trans_norm.weight, dec.pos_trans_norm.bias)
            st["qpos0"].copy_(qp0)
            st["ref0"].copy_(ref_all[:max_len, None, :].expand(max_len, N, 2))
            st["refin0"].copy_(ops.ref_scale_fwd(st["ref0"].view(-1, 2), vr_s.repeat(max_len, 1, 1).contiguous(), 1, geo.L).view(max_len, N, geo.L, 2))
            wsb = st["ws"]
            ops.token_embed_fwd_into(dec.token_embed.weight, toks, deltas, wsb["emb"])
            alive_i32 = st["alive_i32"]
            alive_i32.zero_()

            plan = None
            if mega:
                pkey = (sentinel, self._decode_weights.key)            # by value: the folded weights are rebuilt when a source changes
                if st.get("plan_key") != pkey:
                    st["plan"], st["plan_key"] = self._decode_plan(dec, dw, caches, wsb["emb"], vr_s, geo, N), pkey
                plan = st["plan"]

            def step_body(i, toks_i, deltas_i):
                if toks_i is not toks:                       # teacher forcing: the step's input tokens come from the stream
                    ops.token_embed_fwd_into(dec.token_embed.weight, toks_i, deltas_i, wsb["emb"])
                if plan is not None:
                    plan.launch(i, st["qpos0"][i], st["refin0"][i], st["ref0"][i], out_logits[:, i], out_coords[:, i], out_hs[:, i])
                else:
                    decode_step_fused(dec, dw, wsb, caches, geo, vr_s, i, st["qpos0"][i], st["refin0"][i], st["ref0"][i], out_logits,
                                      out_coords, out_hs)
                if toks_i is toks:
                    ops.decode_advance(out_logits[:, i], out_coords[:, i], unfinished, toks, deltas, i, N, tok.num_bins, min_len,
                                       tok.eos, tok.sep, tok.pad, table=dec.token_embed.weight, embed_out=wsb["emb"],
                                       alive_out=alive_i32[i:i + 1])
            alive_after = alive_i32

        use_graphs = graph and st["calls"] >= 2            # call 1 of a geometry: eager (also the warm-up the capture needs)
        ev_loop0 = ev_loop1 = None
        if timing:                                          # bench.py: the decode loop alone (image encoding excluded)
            ev_loop0, ev_loop1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev_loop0.record()
        i, T = 0, max_len
        while i < max_len:
            if teacher_stream is not None:
                t_i = torch.stack([teacher_stream[k][:, i] for k in ("seq11", "seq12", "seq21", "seq22")]).contiguous()
                d_i = torch.stack([teacher_stream[k][:, i] for k in ("delta_x1", "delta_x2", "delta_y1", "delta_y2")]).contiguous()
                step_body(i, t_i, d_i)
            elif use_graphs:
                g = st["graphs"].get(i)
                if g is None:
                    g = torch.cuda.CUDAGraph()
                    if st["pool"] is None:
                        st["pool"] = torch.cuda.graph_pool_handle()
                    with torch.cuda.graph(g, pool=st["pool"], capture_error_mode="thread_local"):    # DataLoader pin-memory threads may call hipHostMalloc meanwhile
                        step_body(i, toks, deltas)
                    st["graphs"][i] = g
                g.replay()
            else:
                step_body(i, toks, deltas)
            i += 1
            if teacher_stream is None and (i % sync_every == 0 or i == max_len):
                alive = alive_after[:i].cpu()
                done = (alive == 0).nonzero()
                if len(done):
                    T = int(done[0]) + 1
                    break
        else:
            T = i
        T = min(T, i)
        if timing:
            ev_loop1.record()
        incomplete = int(unfinished.sum()) if teacher_stream is None else 0
        if incomplete > 0 and os.environ.get("WARN_INCOMPLETE_GENERATION", "1") == "1":
            warnings.warn(f"{incomplete}/{N} sequences reached max_len={max_len} without predicting EOS.")
        # the state buffers are reused by the next call: hand out copies
        out = {"pred_logits": out_logits[:, :T].clone(), "pred_coords": out_coords[:, :T].clone(), "gen_out": None}
        if timing:
            out["_timing"] = {"events": (ev_loop0, ev_loop1), "steps_run": i, "launch": "graph" if use_graphs else "eager", "fused": fused,
                              "whole_step_kernel": bool(fused and mega)}
        if self.room_class_embed is not None:
            hs2 = out_hs[:, :T].contiguous()
            rl = torch.empty(N, T, self.room_class_embed.weight.shape[0], device=dev)
            ops.gemm(hs2.view(N * T, 256), self.room_class_embed.weight, rl, N * T, rl.shape[-1], 256,
                     bias=self.room_class_embed.bias)
            out["pred_room_logits"] = rl
            out["anchors"] = self.query_embed.weight.detach()
        return out

    def _setup_caches(self, max_bs, max_src_len):
        self.transformer._setup_caches(max_bs, self.seq_len, max_src_len, self.transformer.d_model, self.transformer.nhead,
                                       self.transformer.level_embed.dtype, device=self.transformer.level_embed.device)


def _nearest_mask(mask, h, w):
    iy = torch.arange(h, device=mask.device) * mask.shape[1] // h
    ix = torch.arange(w, device=mask.device) * mask.shape[2] // w
    return mask[:, iy][:, :, ix]


def build(args, train=True, tokenizer=None, cape_mode=False):
    num_classes = 3 if not args.add_cls_token else 4
    pad_idx = tokenizer.pad if tokenizer is not None else 0
    backbone = build_backbone(args)
    transformer = build_deforamble_transformer(args, pad_idx=pad_idx)
    if getattr(args, "model_version", "v1") != "v1":
        raise ValueError("only model_version v1 (RoomFormerV2) is on the CAPE path")
    model = RoomFormerV2(backbone, transformer, num_classes=num_classes, num_queries=args.num_queries,
                         num_polys=args.num_polys, num_feature_levels=args.num_feature_levels, aux_loss=args.aux_loss,
                         with_poly_refine=args.with_poly_refine, masked_attn=args.masked_attn,
                         semantic_classes=args.semantic_classes, seq_len=args.seq_len, tokenizer=tokenizer,
                         use_anchor=args.use_anchor, patch_size=[1, 2][args.image_size == 512],
                         freeze_anchor=getattr(args, "freeze_anchor", False),
                         inject_cls_embed=getattr(args, "inject_cls_embed", False), cape_mode=cape_mode)
    if not train:
        return model
    from .cape_losses import build_cape_criterion
    criterion = build_cape_criterion(args, num_classes=num_classes)
    return model, criterion
