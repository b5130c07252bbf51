"""Isolated timing of the fused decoder kernels at bench size (M = 2^21)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pagnerf_amd import ops, _lib as L
dev = torch.device("cuda:0")
M = 1 << 21
torch.manual_seed(0)
def run(dims, act, out_dtype, grouped, reps=5):
    W = [torch.randn(dims[i + 1], dims[i], device=dev) / dims[i] ** 0.5 for i in range(len(dims) - 1)]
    b = [torch.zeros(dims[i + 1], device=dev) for i in range(len(dims) - 1)]
    for t in W + b: t.requires_grad_(True)
    if grouped:
        x = torch.randn(8, M, 8, device=dev).bfloat16().requires_grad_(True)
    else:
        x = torch.randn(M, dims[0], device=dev).bfloat16().requires_grad_(True)
    g = torch.randn(M, dims[-1], device=dev).to(out_dtype)
    res = {}
    for _ in range(reps):
        ops.profile_start()
        y = ops.fused_mlp(x, W, b, in_dim=dims[0], out_act=act, out_dtype=out_dtype, x1_grouped=(24, 2) if grouped else None)
        y.backward(g)
        prof = ops.profile_stop()
        for k, v in prof.items(): res.setdefault(k, []).append(sum(v))
    print(dims, "act", act, out_dtype, "grouped" if grouped else "rowmajor", {k.replace("pag_", ""): round(min(v), 3) for k, v in res.items()})
for od in (torch.float32, torch.bfloat16):
    run((48, 64, 64, 200), L.ACT_SOFTMAX, od, True)
    run((48, 64, 6), L.ACT_SOFTMAX, od, True)
run((48, 64, 16), L.ACT_NONE, torch.bfloat16, True)
run((48, 64, 16), L.ACT_NONE, torch.bfloat16, False)
