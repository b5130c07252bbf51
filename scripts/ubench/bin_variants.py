"""Time the binned encode backward with experiment builds of encode.hip (scripts/ubench/variants/*.so)."""
import ctypes, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import pagnerf_amd
from pagnerf_amd import _lib as L
dev = torch.device("cuda:0")
M, Lv, F, cap = 4096 * 512, 24, 2, 2 ** 18
# bench-like sample distribution: rays marching through the cube
g = torch.Generator().manual_seed(1)
o = torch.cat([(torch.rand(4096, 2, generator=g) - 0.5) * 0.6, torch.full((4096, 1), 0.95)], 1)
d = torch.nn.functional.normalize(torch.cat([(torch.rand(4096, 2, generator=g) - 0.5) * 0.7, -torch.ones(4096, 1)], 1), dim=-1)
t = ((torch.linspace(0, 1, 512)[None] + torch.rand(4096, 512, generator=g) / 512) ** 2) * 1.9
xyz = (o[:, None] + d[:, None] * t[..., None]).reshape(-1, 3).contiguous().to(dev)
sf = pagnerf_amd.PermutoGridHIP.scale_factors(np.geomspace(1.0, 1e-4, Lv))
shift = torch.randn(Lv, 3, generator=g) * 10
sfh, shh = L.host_floats(sf), L.host_floats(shift)
go = torch.randn(8, M, 8, device=dev).bfloat16()
gt = torch.zeros(Lv, cap, F, device=dev)
for path in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "variants", "*.so"))):
    lib = ctypes.CDLL(path)
    lib.pag_encode_bwd_workspace_bytes.restype = ctypes.c_int64
    lib.pag_encode_bwd_workspace_bytes.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64]
    fn = lib.pag_permuto_encode_bwd
    fn.restype = ctypes.c_int
    fn.argtypes = L._SIGS["pag_permuto_encode_bwd"][1]
    wsb = lib.pag_encode_bwd_workspace_bytes(M, Lv, F, 4, cap)
    ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
    ts = []
    for rep in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc = fn(xyz.data_ptr(), M, go.data_ptr(), L.BF16, 0, 0, L.LAYOUT_XCD8, Lv, F, cap, sfh, shh, None, gt.data_ptr(), ws.data_ptr(), wsb, None)
        b.record(); torch.cuda.synchronize()
        assert rc == 0
        ts.append(a.elapsed_time(b))
    print(os.path.basename(path), "bin+reduce ms:", round(min(ts), 3))
