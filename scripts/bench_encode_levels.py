"""Where the permutohedral forward's time goes per level and per XCD group (kernel experiments):
   python scripts/bench_encode_levels.py      prints (a) the time with all 24 levels at ONE scale, per scale rank, (b) alternative level -> XCD-group assignments
   emulated by permuting the scale list (group g encodes levels g, g+8, g+16 of the list it is given)."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pagnerf_amd import ops, grids  # noqa: E402

dev = torch.device("cuda:0")
N, S = 4096, 512
rays, _ = bench.make_rays(N, dev, seed=1000)
torch.manual_seed(0)
xyz = ops.raymarch_ray(rays.origins, rays.dirs, 0.0, 1.9, S, torch.rand(N, S, device=dev), None, 7)[2].contiguous()
M = xyz.shape[0]
L, F, cap = 24, 2, 1 << 18
scales = np.geomspace(1.0, 1e-4, L)
shift = torch.randn(L, 3) * 10
tab = torch.randn(L, cap, F, device=dev) * 1e-2


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


def run(sc):
    spec = ops.permuto_spec(grids.PermutoGridHIP.scale_factors(np.asarray(sc)), shift, cap, F, half_coords=True)
    with torch.no_grad():
        return timeit(lambda: ops.encode(xyz, tab, spec, None, torch.bfloat16, layout="xcd8"))


print("baseline order            %.4f ms" % run(scales))
for r in (0, 4, 8, 10, 12, 14, 16, 18, 20, 23):
    print("all 24 levels at rank %2d (scale %.2e): %.4f ms  = %.1f us per level" % (r, scales[r], run([scales[r]] * L), 1e3 * run([scales[r]] * L) / L))
perms = {
    "mid band reversed (g, 15-g, 16+g)": list(range(8)) + list(range(15, 7, -1)) + list(range(16, 24)),
    "fine band reversed (g, 8+g, 23-g)": list(range(16)) + list(range(23, 15, -1)),
    "coarse+mid reversed": list(range(7, -1, -1)) + list(range(15, 7, -1)) + list(range(16, 24)),
    "interleaved thirds (3g, 3g+1, 3g+2)": [3 * g for g in range(8)] + [3 * g + 1 for g in range(8)] + [3 * g + 2 for g in range(8)],
}
perms.update({
    "mirror of mid-reversed (7-g, 8+g, 23-g)": list(range(7, -1, -1)) + list(range(8, 16)) + list(range(23, 15, -1)),
    "fine reversed only, again": list(range(16)) + list(range(23, 15, -1)),
    "mid band reversed, again": list(range(8)) + list(range(15, 7, -1)) + list(range(16, 24)),
    "fine band rotated by 4": list(range(16)) + [16 + (j + 4) % 8 for j in range(8)],
    "mid rotated by 4": list(range(8)) + [8 + (j + 4) % 8 for j in range(8)] + list(range(16, 24)),
    "mid reversed + coarse reversed": list(range(7, -1, -1)) + list(range(15, 7, -1)) + list(range(16, 24)),
    "identity again": list(range(24)),
})
if len(sys.argv) > 1:
    perms = {k: v for k, v in perms.items() if sys.argv[1] in k}
for name, p in perms.items():
    print("%-40s %.4f ms" % (name, run(scales[p])))
