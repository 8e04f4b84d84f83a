"""AdamW with global-norm gradient clipping over flat arenas, every scalar on the device
(`cape_sumsq` + `cape_adamw_step`).  Semantics of `torch.optim.AdamW(param_dicts, lr, weight_decay)` after
`torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)` (reference
`train_cape_episodic.py:527-538`, `engine_cape.py:240-258`).  Exposes `param_groups` (lr schedulers work),
`step`, `zero_grad`, and `state_dict`/`load_state_dict` in the per-parameter format *and id enumeration* of the reference's
optimizer, so optimizer checkpoints interchange with `torch.optim.AdamW(param_dicts)` built as in
`train_cape_episodic.py:527-538`: group 0 = every trainable tensor whose name lacks "backbone" in `named_parameters`
order -- including the 38 tensors that never receive a gradient, which hold an id but no state --
group 1 = the backbone tensors."""
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
        # reference enumeration: (group, position) -> arena slot or None (a never-trained tensor)
        slot = {id(p): (ai, i) for ai, a in enumerate(self.arenas) for i, p in enumerate(a.params)}
        self._layout = [[], []]
        seen = set()
        for n, p in model.named_parameters():
            if not p.requires_grad or id(p) in seen:
                continue
            seen.add(id(p))
            self._layout[1 if "backbone" in n else 0].append((n, tuple(p.shape), slot.get(id(p))))
        groups = [{"params": self.arenas[0].params, "lr": lr}, {"params": self.arenas[1].params, "lr": lr_backbone}]
        super().__init__(groups, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.step_count = torch.zeros(1, dtype=torch.int64, device=device)
        # learning rates as device scalars (one per group): the AdamW kernel reads them there, so a step captured into a hipGraph
        # keeps following the schedule -- `sync_lr` uploads param_groups[i]["lr"] when a scheduler changed it
        self.lr_dev = torch.tensor([lr, lr_backbone], dtype=torch.float32, device=device)
        self._lr_host = [lr, lr_backbone]
        # per-block partial sums of squares of the two gradient arenas (no atomics: replicas compute identical clip coefficients)
        from ..hip import lib as _lib
        self._parts = _lib.SUMSQ_PARTS
        self.sumsq = torch.zeros(2 * self._parts, dtype=torch.float32, device=device)
        # weight-gradient kernels may now accumulate straight into the gradient arenas (hip/functional.py)
        HF.Runtime.direct_grad = device.type == "cuda"

    def sync_lr(self):
        """Upload the groups' learning rates if a scheduler moved them (never inside a capture: call before a replay)."""
        cur = [float(g["lr"]) for g in self.param_groups]
        if cur != self._lr_host and not (self.lr_dev.is_cuda and torch.cuda.is_current_stream_capturing()):
            self.lr_dev.copy_(torch.tensor(cur, dtype=torch.float32), non_blocking=False)
            self._lr_host = cur

    @torch.no_grad()
    def step(self, closure=None):
        self.sync_lr()
        HF.Runtime.join()                       # side-stream wgrad kernels must have landed in the arenas
        if self.max_norm > 0:
            for i, a in enumerate(self.arenas):
                if a.numel:
                    ops.sumsq(a.grad, self.sumsq[i * self._parts:(i + 1) * self._parts])
        ops.step_increment(self.step_count)
        for gi, (a, g) in enumerate(zip(self.arenas, self.param_groups)):
            if a.numel:
                b1, b2 = g["betas"]
                ops.adamw_step(a.data, a.grad, a.exp_avg, a.exp_avg_sq, g["lr"], b1, b2, g["eps"], g["weight_decay"],
                               self.max_norm, self.sumsq, self.step_count, lr_dev=self.lr_dev[gi:gi + 1] if self.lr_dev.is_cuda else None)
        # the kernel above rewrote every weight behind autograd's version counters: re-pack the MFMA-fragment copies the
        # register-stationary GEMM reads (one launch over the registered table)
        ops.PackedWeights.invalidate_and_repack()

    def zero_grad(self, set_to_none=False):
        for a in self.arenas:
            a.zero_grad()
        for _, p in self.dead:
            p.grad = None

    def grad_norm(self):
        """Global gradient norm of the last `step` (device tensor, no sync)."""
        return self.sumsq.sum().sqrt()

    # ---- torch-format state dicts, ids as the reference's optimizer assigns them ------------------------------
    def state_dict(self):
        state, groups, idx = {}, [], 0
        for layout, g in zip(self._layout, self.param_groups):
            ids = []
            for _, _, sl in layout:
                if sl is not None:
                    a = self.arenas[sl[0]]
                    p, o = a.params[sl[1]], a.offsets[sl[1]]
                    state[idx] = {"step": self.step_count.clone().float().reshape(()),
                                  "exp_avg": a._view(a.exp_avg, p, o).clone(), "exp_avg_sq": a._view(a.exp_avg_sq, p, o).clone()}
                ids.append(idx)
                idx += 1
            groups.append({k: v for k, v in g.items() if k != "params"} | {"params": ids})
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        """Validates the whole dict (group sizes, shapes) before the first copy: a mismatching checkpoint leaves the
        moments untouched."""
        sgroups = sd["param_groups"]
        if len(sgroups) != len(self._layout):
            raise ValueError(f"optimizer state has {len(sgroups)} param groups, expected {len(self._layout)}")
        plan = []
        for gi, (layout, sg) in enumerate(zip(self._layout, sgroups)):
            if len(sg["params"]) != len(layout):
                raise ValueError(f"param group {gi}: {len(sg['params'])} parameters in the state dict, {len(layout)} in the model "
                                 "(the reference keeps its never-trained tensors in group 0)")
            for pid, (name, shape, sl) in zip(sg["params"], layout):
                st = sd["state"].get(pid)
                if st is None:
                    continue
                if sl is None:
                    raise ValueError(f"state for {name}, which never receives a gradient on the CAPE path")
                for k in ("exp_avg", "exp_avg_sq"):
                    if tuple(st[k].shape) != shape:
                        raise ValueError(f"{name}.{k}: shape {tuple(st[k].shape)} in the state dict, {shape} in the model")
                plan.append((sl, st))
        for g, sg in zip(self.param_groups, sgroups):
            for k, v in sg.items():
                if k != "params":
                    g[k] = v
        for sl, st in plan:
            a = self.arenas[sl[0]]
            p, o = a.params[sl[1]], a.offsets[sl[1]]
            a._view(a.exp_avg, p, o).copy_(st["exp_avg"])
            a._view(a.exp_avg_sq, p, o).copy_(st["exp_avg_sq"])
            self.step_count.fill_(int(float(st["step"])))
