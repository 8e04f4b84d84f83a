"""CAPE criterion (reference `models/cape_losses.py:39-202`, `roomformer_v2.py:915-953`): class-weighted
cross-entropy over (label != -1) & visible tokens and L1 over visible <coord> tokens, for the last and the
five auxiliary decoder layers.  All layers are evaluated -- values and gradients -- by one HIP kernel
(`cape_loss_fwd_bwd`); this class only mirrors the reference's dict-of-losses interface."""
import torch
import torch.nn as nn

from ..hip import functional as HF
from ..hip import ops


class CAPESetCriterion(nn.Module):
    def __init__(self, num_classes, semantic_classes, matcher, weight_dict, losses, label_smoothing=0.0,
                 per_token_sem_loss=False, eos_weight=20.0):
        super().__init__()
        if label_smoothing != 0.0 or num_classes != 3:
            raise ValueError("CAPE path: label_smoothing=0 and 3 token classes (reference defaults)")
        self.num_classes, self.semantic_classes, self.matcher = num_classes, semantic_classes, matcher
        self.weight_dict, self.losses = weight_dict, losses
        self.label_smoothing, self.per_token_sem_loss = label_smoothing, per_token_sem_loss
        self.eos_weight = eos_weight
        cw = torch.ones(num_classes, dtype=torch.float32)
        cw[2] = eos_weight
        self.register_buffer("class_weights", cw, persistent=False)

    def forward(self, outputs, targets):
        """Returns the 19-entry loss dict of the reference; `loss_dict['_total']` additionally carries the
        weighted sum as a differentiable scalar (what the engine back-propagates)."""
        if "_stack_logits" in outputs:
            logits, coords = outputs["_stack_logits"], outputs["_stack_coords"]
        else:
            aux = outputs.get("aux_outputs", [])
            logits = torch.stack([a["pred_logits"] for a in aux] + [outputs["pred_logits"]])
            coords = torch.stack([a["pred_coords"] for a in aux] + [outputs["pred_coords"]])
        NL = logits.shape[0]
        dev = logits.device
        labels = targets["token_labels"].to(dev)
        vis = ops.as_u8(targets["visibility_mask"].to(dev)) if "visibility_mask" in targets else \
            torch.ones_like(labels, dtype=torch.uint8)
        w_ce, w_l1 = float(self.weight_dict["loss_ce"]), float(self.weight_dict["loss_coords"])
        total, per = HF.cape_loss(logits, coords, labels, vis, targets["target_seq"].to(dev),
                                  self.class_weights.to(dev), w_ce, w_l1)
        names = [f"_{i}" for i in range(NL - 1)] + [""]
        losses = {}
        for i, s in enumerate(names):
            losses["loss_ce" + s] = per[2 * i]
            losses["loss_coords" + s] = per[2 * i + 1]
            losses["cardinality_error" + s] = 0.0
        if "pred_room_logits" in outputs:
            losses["loss_ce_room"] = torch.zeros((), device=dev)      # no <cls> labels on the CAPE path
        losses["_total"] = total
        return losses


def weighted_total(loss_dict, weight_dict):
    """`sum(loss_dict[k] * weight_dict[k])` of engine_cape.py:195, served by the fused kernel's total."""
    return loss_dict["_total"]


def build_cape_criterion(args, num_classes=3):
    weight_dict = {"loss_ce": args.cls_loss_coef, "loss_ce_room": getattr(args, "room_cls_loss_coef", 0.0),
                   "loss_coords": args.coords_loss_coef}
    if getattr(args, "raster_loss_coef", 0) > 0:
        raise ValueError("rasterisation loss is not on the CAPE path (raster_loss_coef must be 0)")
    weight_dict["loss_dir"] = 1
    weight_dict.update({k + "_enc": v for k, v in list(weight_dict.items())})
    if args.aux_loss:
        aux = {}
        for i in range(args.dec_layers - 1):
            aux.update({k + f"_{i}": v for k, v in weight_dict.items()})
        aux.update({k + "_enc": v for k, v in weight_dict.items()})
        weight_dict.update(aux)
    return CAPESetCriterion(num_classes=num_classes, semantic_classes=getattr(args, "semantic_classes", -1), matcher=None,
                            weight_dict=weight_dict, losses=["labels", "polys", "cardinality"],
                            label_smoothing=getattr(args, "label_smoothing", 0.0),
                            per_token_sem_loss=getattr(args, "per_token_sem_loss", False),
                            eos_weight=getattr(args, "eos_weight", 20.0))
