"""Adam on the gfx950 kernel (pag_adam_step) behind torch.optim.Adam's own interface.

The reference builds its optimiser as `optim_cls(params, **optim_params)` with `optim_cls = torch.optim.Adam`, `eps = 1e-15`
(config_parser.py:667-673, `str2optim`) over the parameter groups of pc_nerf/trainer.py:268-286 and steps it through
`self.scaler.step(self.optimizer)` (:583).  `pagnerf_amd.optim.Adam` IS a `torch.optim.Adam` (subclass: same constructor, same
`param_groups`, same `state_dict()` layout - `step` / `exp_avg` / `exp_avg_sq` per parameter -, so LR schedulers, GradScaler and
checkpoints of either class load into the other) whose `step()` hands every fp32 GPU parameter of a group to ONE C-ABI call: the
two 50.3 MB tables as streaming launches, the ~22 decoder tensors in one small launch.  Same arithmetic as torch's single-tensor
formula (csrc/optim.hip).  Why: the update is 704 MB of pure streaming per step whatever the batch, and torch's fused multi-tensor
kernel moves it at 3.6 TB/s on MI355X - 0.196 ms per step, 5 % of the full step and 16 % of the post-prune step.

Anything this kernel does not cover takes torch's own implementation for that group, silently and correctly: amsgrad, maximize,
capturable / differentiable, sparse gradients, CPU tensors, non-fp32 parameters (fp16 tables), non-contiguous tensors."""
import ctypes

import torch

from . import _lib as L
from . import ops


class Adam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, **kwargs):
        kwargs.pop("fused", None)         # our launches replace torch's fused / foreach kernels for the groups they cover
        kwargs.pop("foreach", None)
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, **kwargs)

    @staticmethod
    def _eligible(group, params):
        if group.get("amsgrad") or group.get("maximize") or group.get("capturable") or group.get("differentiable"):
            return False
        if isinstance(group["lr"], torch.Tensor):
            return False
        for p in params:
            g = p.grad
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and g is not None and not g.is_sparse
                    and g.dtype == torch.float32 and g.is_contiguous() and g.device == p.device):
                return False
        return True

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        fallback = []
        for gi, group in enumerate(self.param_groups):
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            if not self._eligible(group, params):
                fallback.append(gi)
                continue
            steps = set()
            for p in params:
                st = self.state[p]
                if len(st) == 0:          # torch's own lazy initialisation (_init_group): step as a CPU scalar tensor, zero moments
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                steps.add(float(st["step"]))
            beta1, beta2 = group["betas"]
            # parameters that joined the optimiser at different times carry different step counts: one call per count
            for step in sorted(steps):
                part = [p for p in params if float(self.state[p]["step"]) == step]
                self._launch(gi, part, float(group["lr"]), float(beta1), float(beta2), float(group["eps"]), float(group["weight_decay"]), int(step))
        if fallback:
            self._torch_step(fallback)
        return loss

    def _launch(self, gi, params, lr, beta1, beta2, eps, weight_decay, step):
        n = len(params)
        arr = lambda vals: (ctypes.c_void_p * n)(*vals)
        pp, gg = arr([p.data_ptr() for p in params]), arr([p.grad.data_ptr() for p in params])
        mm = arr([self.state[p]["exp_avg"].data_ptr() for p in params])
        vv = arr([self.state[p]["exp_avg_sq"].data_ptr() for p in params])
        nn = (ctypes.c_int64 * n)(*[p.numel() for p in params])
        with torch.cuda.device(params[0].device):
            ops._call("pag_adam_step", n, pp, gg, mm, vv, nn, lr, beta1, beta2, eps, weight_decay, step, L.stream())

    def _torch_step(self, group_ids):
        """torch.optim.Adam.step on the groups our kernel does not cover (the others are hidden from it for the call)."""
        keep = self.param_groups
        try:
            self.param_groups = [keep[i] for i in group_ids]
            super().step()
        finally:
            self.param_groups = keep
