"""Time the encode forward / position-gradient / table-gradient launches alone (HIP events), for kernel experiments:
   python scripts/bench_encode.py [M_log2]"""
import sys
import os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pagnerf_amd import ops, grids  # noqa: E402

dev = torch.device("cuda:0")
M = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 21)
L, F, cap = (int(sys.argv[2]) if len(sys.argv) > 2 else 24), 2, 1 << 18
torch.manual_seed(0)
# ray-like sample order: 512 samples along each of M/512 rays
o = (torch.rand(M // 512, 1, 3, device=dev) - 0.5) * 0.2
d = torch.nn.functional.normalize(torch.randn(M // 512, 1, 3, device=dev), dim=-1)
t = torch.linspace(0, 1, 512, device=dev)[None, :, None] ** 2 * 0.9
xyz = (o + d * t).reshape(-1, 3).contiguous()
sf = grids.PermutoGridHIP.scale_factors(np.geomspace(1.0, 1e-4, L))
shift = torch.randn(L, 3) * 10
spec = ops.permuto_spec(sf, shift, cap, F)
tab = (torch.randn(L, cap, F, device=dev) * 1e-2).requires_grad_(True)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


with torch.no_grad():
    print("fwd xcd8 bf16   %.4f ms" % timeit(lambda: ops.encode(xyz, tab, spec, None, torch.bfloat16, layout="xcd8")))
    print("fwd [M,C] f32   %.4f ms" % timeit(lambda: ops.encode(xyz, tab, spec, None, torch.float32)))
out = ops.encode(xyz, tab, spec, None, torch.bfloat16, layout="xcd8")
g = torch.randn_like(out.float()).bfloat16()
print("bwd tables      %.4f ms" % timeit(lambda: torch.autograd.grad(out, tab, g, retain_graph=True)))
x2 = xyz.clone().requires_grad_(True)
out2 = ops.encode(x2, tab.detach(), spec, None, torch.bfloat16, layout="xcd8")
print("bwd xyz         %.4f ms" % timeit(lambda: torch.autograd.grad(out2, x2, g, retain_graph=True)))
