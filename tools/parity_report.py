"""Max deviations of the HIP product from the reference's golden vectors, per GEMM precision mode."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cape_ref, procweights, synth  # noqa: E402
from tests.helpers import build_product, to_dev  # noqa: E402
from cape_amd.hip import ops  # noqa: E402

CFG = cape_ref.Cfg()
sd = procweights.procedural_state_dict()
G = os.path.join(ROOT, "tests", "golden")


def run(prec):
    ops.set_gemm_precision(prec)
    res = {}
    args, tok, model, crit = build_product(proc_sd=sd)
    model.eval()
    for tag, mk, sl in (("e2e64", lambda: synth.make_batch(11, 2, 2, 64, 9, CFG, n_invisible=(2, 0)), slice(None)),
                        ("e2e256", lambda: synth.make_batch(23, 1, 2, 256, 17, CFG, n_invisible=(2,)), slice(0, 24))):
        d = np.load(os.path.join(G, tag + ".npz"))
        b = to_dev(mk())
        model.zero_grad(set_to_none=True)
        out = model(samples=b["images"], support_coords=b["support_coords"], support_mask=b["support_mask"],
                    targets=b["targets"], skeleton_edges=b["skeleton"])
        logits = torch.stack([a["pred_logits"] for a in out["aux_outputs"]] + [out["pred_logits"]])[:, :, sl].cpu()
        coords = torch.stack([a["pred_coords"] for a in out["aux_outputs"]] + [out["pred_coords"]])[:, :, sl].cpu()
        rl, rc = torch.from_numpy(d["logits"]), torch.from_numpy(d["coords"])
        res[tag] = {"max_dlogit": float((logits - rl).abs().max()), "max_dcoord": float((coords - rc).abs().max()),
                    "argmax_equal": bool(torch.equal(logits.argmax(-1), rl.argmax(-1)))}
        if tag == "e2e64":
            tot = crit(out, b["targets"])["_total"]
            tot.backward()
            named = dict(model.named_parameters(remove_duplicate=False))
            gk = json.loads(bytes(d["gnorm_keys"]).decode())
            worst = max(abs(float(named[n].grad.norm()) - r) / max(r, 1e-3) for n, r in zip(gk, d["gnorm_vals"]))
            res[tag]["loss_abs_err"] = abs(float(tot) - float(d["loss"]))
            res[tag]["worst_rel_gradnorm_err"] = worst
    return res


if __name__ == "__main__":
    out = {p: run(p) for p in ("f32", "bf16x3")}
    print(json.dumps(out, indent=1))
