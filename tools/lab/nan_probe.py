"""Where does a non-finite value first appear in a training step after decode graphs were captured? (lab)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
if os.environ.get("POISON", "0") == "1":
    _empty, _empty_like = torch.empty, torch.empty_like
    def _pe(*a, **k):
        t = _empty(*a, **k)
        return t.fill_(float("nan")) if t.is_floating_point() else t
    def _pel(*a, **k):
        t = _empty_like(*a, **k)
        return t.fill_(float("nan")) if t.is_floating_point() else t
    torch.empty, torch.empty_like = _pe, _pel
from test_configs_gpu import build_product
from cape_amd.runtime.optimizer import ArenaAdamW
from cape_amd.hip import functional as HF
from cape_amd.datasets.synthetic import SyntheticEpisodes
from cape_amd.datasets import episodic_collate_fn

pre_decode = os.environ.get("PRE_DECODE", "1") == "1"
args, tok, model, crit = build_product(proc_sd=None)
tok.seq_len = 10
opt = ArenaAdamW(model, lr=3e-3, lr_backbone=3e-4, weight_decay=1e-4, max_norm=0.1)
HF.Runtime.seed(5, "cuda")
g = torch.Generator().manual_seed(4)
imgs = torch.rand(2, 3, 256, 256, generator=g).cuda()
sc = torch.rand(2, 9, 2, generator=g).cuda()
sm = (torch.arange(9)[None, :] < torch.tensor([[6], [7]])).cuda()
sk = [[[0, 1], [1, 2]], [[0, 1], [2, 3]]]
if pre_decode:
    os.environ["CAPE_DECODE_FUSED"] = "1"
    model.eval()
    with torch.no_grad():
        for _ in range(3):
            model.forward_inference(samples=imgs, support_coords=sc, support_mask=sm, skeleton_edges=sk, graph=True)
model.train()
ds = SyntheticEpisodes(tok, 2, 256, 9, 1, seed=3)
bt = episodic_collate_fn([ds[0], ds[1]])
for it in range(int(os.environ.get("ITERS", "2"))):
    out = model(samples=bt["query_images"].cuda(), support_coords=bt["support_coords"].cuda(), support_mask=bt["support_masks"].cuda(),
                targets={k: v.cuda() for k, v in bt["query_targets"].items()}, skeleton_edges=bt["support_skeletons"])
    for k, v in out.items():
        if torch.is_tensor(v):
            print(it, "out", k, bool(torch.isfinite(v).all()))
    loss = crit(out, {k: v.cuda() for k, v in bt["query_targets"].items()})["_total"]
    print(it, "loss", float(loss))
    loss.backward()
    HF.Runtime.join(); torch.cuda.synchronize()
    bad = [(n, int((~torch.isfinite(p.grad)).sum())) for n, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    print(it, "non-finite grads:", bad[:12], len(bad))
    opt.step()
    torch.cuda.synchronize()
    print(it, "sumsq finite", bool(torch.isfinite(opt.sumsq).all()), "norm", float(opt.grad_norm()))
    badp = [n for n, p in model.named_parameters() if not torch.isfinite(p).all()]
    print(it, "non-finite params:", badp[:8], len(badp))
    opt.zero_grad()
