"""CAPEModel: support-graph conditioned wrapper around RoomFormerV2 -- the drop-in boundary
(reference `models/cape_model.py:9-229`): same constructor, `forward`, `forward_inference`,
`build_cape_model`, same `state_dict` keys (including the never-used
`support_cross_attention_layers` / `support_attn_layer_norms`) and the same
support-mask inversion quirk (fact 4).  All arithmetic runs in the HIP kernels of libcape_hip.so."""
import torch
import torch.nn as nn

from .geometric_support_encoder import GeometricSupportEncoder


class CAPEModel(nn.Module):
    def __init__(self, base_model, hidden_dim=256, support_encoder_layers=3, support_fusion_method="cross_attention",
                 use_geometric_encoder=False, use_gcn_preenc=False, num_gcn_layers=2):
        super().__init__()
        self.base_model = base_model
        self.hidden_dim = hidden_dim
        self.support_fusion_method = support_fusion_method
        self.use_geometric_encoder = use_geometric_encoder
        if not use_geometric_encoder:
            raise ValueError("cape_amd implements the geometric support encoder (--use_geometric_encoder); the legacy "
                             "SupportPoseGraphEncoder is outside the hot path named by the north star")
        self.support_encoder = GeometricSupportEncoder(hidden_dim=hidden_dim, num_encoder_layers=support_encoder_layers,
                                                       nhead=8, dim_feedforward=1024, dropout=0.1,
                                                       use_gcn_preenc=use_gcn_preenc, num_gcn_layers=num_gcn_layers,
                                                       activation="relu")
        if support_fusion_method == "cross_attention":
            self._add_support_cross_attention()
        elif support_fusion_method == "concat":
            self.support_proj = nn.Linear(hidden_dim * 2, hidden_dim)
        elif support_fusion_method == "add":
            pass
        else:
            raise ValueError(f"Unknown fusion method: {support_fusion_method}")

    def _add_support_cross_attention(self):
        """Parameters that exist in the reference's checkpoints but are never read by the decoder."""
        num_layers = self.base_model.transformer.decoder.num_layers
        self.support_cross_attention_layers = nn.ModuleList(
            [nn.MultiheadAttention(embed_dim=self.hidden_dim, num_heads=8, dropout=0.1, batch_first=True)
             for _ in range(num_layers)])
        self.support_attn_layer_norms = nn.ModuleList([nn.LayerNorm(self.hidden_dim) for _ in range(num_layers)])

    def _check(self, samples, support_coords, support_mask, skeleton_edges):
        if isinstance(samples, torch.Tensor):
            qbs = samples.shape[0]
        elif hasattr(samples, "tensors"):
            qbs = samples.tensors.shape[0]
        else:
            qbs = len(samples)
        sbs = support_coords.shape[0]
        if sbs != qbs:
            raise ValueError("Support-Query batch size mismatch! This breaks 1-shot episodic structure.\n"
                             f"  Support batch size: {sbs}\n  Query batch size: {qbs}\n"
                             "Expected: Both should be (B*K) where B=episodes, K=queries_per_episode.")
        if support_mask.shape[0] != sbs:
            raise ValueError(f"Support mask batch size ({support_mask.shape[0]}) doesn't match support_coords batch size ({sbs})")
        if skeleton_edges is not None and len(skeleton_edges) != sbs:
            raise ValueError(f"Skeleton edges list length ({len(skeleton_edges)}) doesn't match batch size ({sbs})")

    def _inject(self, support_features, support_mask):
        dec = self.base_model.transformer.decoder
        dec.support_features = support_features
        dec.support_mask = support_mask
        dec.support_cross_attn_layers = getattr(self, "support_cross_attention_layers", None)
        dec.support_attn_norms = getattr(self, "support_attn_layer_norms", None)

    def _clear(self):
        dec = self.base_model.transformer.decoder
        dec.support_features = None
        dec.support_mask = None
        dec.support_cross_attn_layers = None
        dec.support_attn_norms = None

    def forward(self, samples, support_coords, support_mask, targets=None, skeleton_edges=None):
        self._check(samples, support_coords, support_mask, skeleton_edges)
        if support_mask.dtype != torch.bool:
            support_mask = support_mask.bool()
        encoder_mask = ~support_mask                      # the inversion of cape_model.py:120-124
        support_features = self.support_encoder(support_coords, encoder_mask, skeleton_edges)
        self._inject(support_features, support_mask)
        try:
            outputs = self.base_model(samples, seq_kwargs=targets)
        finally:
            self._clear()
        return outputs

    def forward_inference(self, samples, support_coords, support_mask, skeleton_edges=None, max_seq_len=None, use_cache=True,
                          teacher_stream=None, graph=None, timing=False):
        """`teacher_stream`, `graph`, `timing` are keyword extras of this implementation (parity tests, hipGraph switch, bench
        events); the reference signature ends at `use_cache`."""
        if support_mask.dtype != torch.bool:
            support_mask = support_mask.bool()
        encoder_mask = ~support_mask
        support_features = self.support_encoder(support_coords, encoder_mask, skeleton_edges)
        self._inject(support_features, support_mask)
        try:
            with torch.no_grad():
                outputs = self.base_model.forward_inference(samples=samples, use_cache=use_cache, teacher_stream=teacher_stream,
                                                             graph=graph, timing=timing)
        finally:
            self._clear()
        pred_logits = outputs.get("pred_logits")
        pred_coords = outputs.get("pred_coords")
        # argmax over 3 classes on the host glue side: (N,T,3) -> (N,T) int64 ('sequences' of the reference API)
        pred_tokens = pred_logits.argmax(dim=-1) if pred_logits is not None else None
        res = {"sequences": pred_tokens, "coordinates": pred_coords, "logits": pred_logits}
        if timing:
            res["_timing"] = outputs.get("_timing")
        return res


def build_cape_model(args, base_model):
    return CAPEModel(base_model=base_model, hidden_dim=getattr(args, "hidden_dim", 256),
                     support_encoder_layers=getattr(args, "support_encoder_layers", 3),
                     support_fusion_method=getattr(args, "support_fusion_method", "cross_attention"),
                     use_geometric_encoder=getattr(args, "use_geometric_encoder", False),
                     use_gcn_preenc=getattr(args, "use_gcn_preenc", False),
                     num_gcn_layers=getattr(args, "num_gcn_layers", 2))
