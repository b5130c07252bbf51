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
capturable / differentiable, decoupled weight decay (AdamW's `p *= 1 - lr * wd`; the kernel implements Adam's L2 form `g += wd * p`),
tensor-valued lr / betas, sparse gradients, CPU tensors, non-fp32 parameters (fp16 tables), non-contiguous tensors.

Host cost matters as much as the kernel in the post-prune regime (a 0.6 ms step): per group the parameter list, the moment pointers and
the step counters are prepared ONCE (`_Plan`) and a step is one pass over the gradients (pointer + dtype / layout check), one
`_foreach_add_` on the step counters and one ctypes call; the plan is rebuilt when the set of parameters that carry a gradient, or any
state tensor, changes (load_state_dict, a parameter frozen / unfrozen)."""
import ctypes

import torch

from . import _lib as L
from . import ops


class _Plan:
    """Everything of a parameter group's step that does not change from step to step."""

    def __init__(self, opt, params):
        self.params = params
        self.ids = tuple(id(p) for p in params)
        for p in params:
            st = opt.state[p]
            if len(st) == 0:          # torch's own lazy initialisation (Adam._init_group): step as a CPU scalar tensor, zero moments
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        states = [opt.state[p] for p in params]
        self.steps = [st["step"] for st in states]
        self.state_ids = tuple((id(st["step"]), id(st["exp_avg"]), id(st["exp_avg_sq"])) for st in states)
        self.ok = all(isinstance(t, torch.Tensor) and not t.is_cuda for t in self.steps) and \
            all(st["exp_avg"].is_cuda and st["exp_avg"].dtype == torch.float32 and st["exp_avg"].is_contiguous() and
                st["exp_avg_sq"].dtype == torch.float32 and st["exp_avg_sq"].is_contiguous() for st in states)
        counts = {float(t) for t in self.steps} if self.ok else set()
        self.uniform = len(counts) == 1          # every parameter has taken the same number of steps (the normal case)
        self.count = int(counts.pop()) if self.uniform else None
        n = len(params)
        self.n = n
        arr = lambda vals: (ctypes.c_void_p * n)(*vals)
        self.pp = arr([p.data_ptr() for p in params])
        self.mm = arr([st["exp_avg"].data_ptr() for st in states])
        self.vv = arr([st["exp_avg_sq"].data_ptr() for st in states])
        self.ptrs = tuple(p.data_ptr() for p in params)
        self.nn = (ctypes.c_int64 * n)(*[p.numel() for p in params])
        self.gg = (ctypes.c_void_p * n)()
        self.device = params[0].device


class Adam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, **kwargs):
        kwargs.pop("fused", None)         # our launches replace torch's fused / foreach kernels for the groups they cover
        kwargs.pop("foreach", None)
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, **kwargs)
        self._plans = {}

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._plans = {}

    @staticmethod
    def _group_ok(group):
        if group.get("amsgrad") or group.get("maximize") or group.get("capturable") or group.get("differentiable"):
            return False
        if group.get("decoupled_weight_decay") and group.get("weight_decay", 0) != 0:       # torch >= 2.7 Adam(decoupled_weight_decay=True) = AdamW
            return False
        return not (isinstance(group["lr"], torch.Tensor) or any(isinstance(b, torch.Tensor) for b in group["betas"]))

    def _plan(self, gi, group):
        """The group's plan, or None when a parameter / gradient is outside what the kernel covers (-> torch's step for the group)."""
        params = [p for p in group["params"] if p.grad is not None]
        if not params:
            return False
        plan = self._plans.get(gi)
        ids = tuple(id(p) for p in params)
        if plan is None or plan.ids != ids or any(p.data_ptr() != q for p, q in zip(params, plan.ptrs)) or \
                any(id(self.state[p].get("exp_avg")) != s[1] for p, s in zip(params, plan.state_ids)):
            if not all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in params):
                return None
            plan = self._plans[gi] = _Plan(self, params)
        if not plan.ok:
            return None
        gg = plan.gg
        f32 = torch.float32
        for i, p in enumerate(params):
            g = p.grad
            if g.dtype is not f32 or g.is_sparse or not g.is_contiguous() or g.device != plan.device:
                return None
            gg[i] = g.data_ptr()
        return plan

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        fallback = []
        for gi, group in enumerate(self.param_groups):
            plan = self._plan(gi, group) if self._group_ok(group) else None
            if plan is False:
                continue
            if plan is None:
                if any(p.grad is not None for p in group["params"]):
                    fallback.append(gi)
                continue
            beta1, beta2 = group["betas"]
            hyper = (float(group["lr"]), float(beta1), float(beta2), float(group["eps"]), float(group["weight_decay"]))
            # the step counters advance only once the launch has been accepted: a call that raises (bad hyper-parameter, launch failure)
            # leaves counters, moments and parameters as they were
            if plan.uniform:
                self._launch(plan, plan.pp, plan.gg, plan.mm, plan.vv, plan.nn, plan.n, hyper, plan.count + 1)
                plan.count += 1
            else:       # parameters that joined the optimiser at different times carry different step counts: one call per count
                by = {}
                for i, t in enumerate(plan.steps):
                    by.setdefault(int(float(t)) + 1, []).append(i)
                for count, idx in sorted(by.items()):
                    n = len(idx)
                    sel = lambda a, ct: (ct * n)(*[a[i] for i in idx])
                    self._launch(plan, sel(plan.pp, ctypes.c_void_p), sel(plan.gg, ctypes.c_void_p), sel(plan.mm, ctypes.c_void_p),
                                 sel(plan.vv, ctypes.c_void_p), sel(plan.nn, ctypes.c_int64), n, hyper, count)
            torch._foreach_add_(plan.steps, 1)
        if fallback:
            self._torch_step(fallback)
        return loss

    @staticmethod
    def _launch(plan, pp, gg, mm, vv, nn, n, hyper, count):
        if plan.device.index is not None and plan.device.index != torch.cuda.current_device():
            with torch.cuda.device(plan.device):
                ops._call("pag_adam_step", n, pp, gg, mm, vv, nn, *hyper, count, L.stream())
        else:
            ops._call("pag_adam_step", n, pp, gg, mm, vv, nn, *hyper, count, L.stream())

    def _torch_step(self, group_ids):
        """torch.optim.Adam.step on the groups our kernel does not cover (the others are hidden from it for the call).  Called from inside
        this class' own (hook-wrapped) step(): torch's step is entered BELOW its hook wrapper (Optimizer.profile_hook_step marks the class'
        `step` as `hooked` once a plain torch.optim.Adam has been constructed), so step pre / post hooks and the profiler record fire once
        per step() of this optimiser, not a second time for the fallback groups - and no hook sees the temporary group list."""
        keep = self.param_groups
        inner = torch.optim.Adam.step
        if getattr(inner, "hooked", False) and hasattr(inner, "__wrapped__"):
            inner = inner.__wrapped__
        try:
            self.param_groups = [keep[i] for i in group_ids]
            inner(self)
        finally:
            self.param_groups = keep
        self._plans = {k: v for k, v in self._plans.items() if k not in group_ids}
