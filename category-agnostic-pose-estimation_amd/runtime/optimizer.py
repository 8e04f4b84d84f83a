"""AdamW with global-norm gradient clipping over flat arenas, every scalar on the device
(`cape_sumsq` + `cape_adamw_step`).  Semantics of `torch.optim.AdamW(param_dicts, lr, weight_decay)` after
`torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)` (reference
`train_cape_episodic.py:527-538`, `engine_cape.py:240-258`).  Exposes `param_groups` (lr schedulers work),
`step`, `zero_grad`, `state_dict`/`load_state_dict` in torch's per-parameter format so checkpoints interchange."""
import torch

from ..hip import functional as HF
from ..hip import ops
from .arena import ParamGroupArena, split_groups


class ArenaAdamW(torch.optim.Optimizer):
    def __init__(self, model, lr=1e-4, lr_backbone=1e-5, weight_decay=1e-4, betas=(0.9, 0.999), eps=1e-8, max_norm=0.0):
        device = next(model.parameters()).device
        main, backbone, dead = split_groups(model)
        self.arenas = [ParamGroupArena(main, device), ParamGroupArena(backbone, device)]
        self.dead = dead
        self.max_norm = max_norm
        groups = [{"params": self.arenas[0].params, "lr": lr}, {"params": self.arenas[1].params, "lr": lr_backbone}]
        super().__init__(groups, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.step_count = torch.zeros(1, dtype=torch.int64, device=device)
        self.sumsq = torch.zeros(1, dtype=torch.float32, device=device)
        # weight-gradient kernels may now accumulate straight into the gradient arenas (hip/functional.py)
        HF.Runtime.direct_grad = device.type == "cuda"

    @torch.no_grad()
    def step(self, closure=None):
        HF.Runtime.join()                       # side-stream wgrad kernels must have landed in the arenas
        self.sumsq.zero_()
        if self.max_norm > 0:
            for a in self.arenas:
                if a.numel:
                    ops.sumsq(a.grad, self.sumsq)
        ops.step_increment(self.step_count)
        for a, g in zip(self.arenas, self.param_groups):
            if a.numel:
                b1, b2 = g["betas"]
                ops.adamw_step(a.data, a.grad, a.exp_avg, a.exp_avg_sq, g["lr"], b1, b2, g["eps"], g["weight_decay"],
                               self.max_norm, self.sumsq, self.step_count)

    def zero_grad(self, set_to_none=False):
        for a in self.arenas:
            a.zero_grad()
        for _, p in self.dead:
            p.grad = None

    def grad_norm(self):
        """Global gradient norm of the last `step` (device tensor, no sync)."""
        return self.sumsq.sqrt()

    # ---- torch-format state dicts -------------------------------------------------------------
    def state_dict(self):
        state, idx = {}, 0
        groups = []
        for a, g in zip(self.arenas, self.param_groups):
            ids = []
            for p, o in zip(a.params, a.offsets):
                state[idx] = {"step": self.step_count.clone().float().reshape(()),
                              "exp_avg": a._view(a.exp_avg, p, o).clone(), "exp_avg_sq": a._view(a.exp_avg_sq, p, o).clone()}
                ids.append(idx)
                idx += 1
            groups.append({k: v for k, v in g.items() if k != "params"} | {"params": ids})
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        idx = 0
        for a, g, sg in zip(self.arenas, self.param_groups, sd["param_groups"]):
            for k, v in sg.items():
                if k != "params":
                    g[k] = v
            for p, o in zip(a.params, a.offsets):
                st = sd["state"].get(idx)
                if st is not None:
                    a._view(a.exp_avg, p, o).copy_(st["exp_avg"])
                    a._view(a.exp_avg_sq, p, o).copy_(st["exp_avg_sq"])
                    self.step_count.fill_(int(float(st["step"])))
                idx += 1
