"""Time the permutohedral encode forward (production layout: bf16 [8,M,8], fp16-rounded coordinates) on the bench's own samples:
   [PAG_LIB_VARIANT=tag] python scripts/bench_encode_fwd.py [rays] [samples]      (kernel experiments: scripts/build_variant.sh)"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pagnerf_amd import ops, grids  # noqa: E402

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
S = int(sys.argv[2]) if len(sys.argv) > 2 else 512
tdt = torch.float16 if (len(sys.argv) > 3 and sys.argv[3] == "fp16") else torch.float32
rays, _ = bench.make_rays(N, dev, seed=1000)
torch.manual_seed(0)
out = ops.raymarch_ray(rays.origins, rays.dirs, 0.0, 1.9, S, torch.rand(N, S, device=dev), None, 7)
xyz = out[2].contiguous()
M = xyz.shape[0]
L, F, cap = 24, 2, 1 << 18
sf = grids.PermutoGridHIP.scale_factors(np.geomspace(1.0, 1e-4, L))
shift = torch.randn(L, 3) * 10
spec = ops.permuto_spec(sf, shift, cap, F, half_coords=True)
tab = (torch.randn(L, cap, F, device=dev) * 1e-2).to(tdt)
other = torch.randn(8, M, 8, device=dev).bfloat16()


def timeit(fn, n=30):
    for _ in range(5):
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
    t1 = timeit(lambda: ops.encode(xyz, tab, spec, None, torch.bfloat16, layout="xcd8"))
    t2 = timeit(lambda: ops.encode(xyz, tab, spec, None, torch.bfloat16, layout="xcd8", addend=other))
tb = 2 if tdt == torch.float16 else 4
bps = 12 + L * 4 * F * tb + L * F * 2
print("variant %-8s M %d  fwd %.4f ms (%.3f of 8 TB/s)   fwd_add %.4f ms" % (os.environ.get("PAG_LIB_VARIANT", "-") + ("/generic" if os.environ.get("PAG_NO_FAST_ENCODE") else ""),
                                                                           M, t1, bps * M / t1 / 1e6 / 8000.0, t2))
