"""Experiment: the forward encode with d feat / d xyz written next to the features (PAG_EXP_JAC = bf16 values per (sample, XCD group): 24 = 48 B).
    bash scripts/build_variant.sh jac24 encode "-DPAG_EXP_JAC=24"
    PAG_LIB_VARIANT=jac24 python3 scripts/exp_jac_store.py        (and without the variable: the regular library)"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pagnerf_amd import _lib, ops

dev = torch.device("cuda:0")
args = bench.parse(["--graphs", "off"])
nef = bench.make_model(args, dev, 0)
g = nef.grid
lib = _lib.load()
per = int(os.environ.get("PAG_EXP_JAC_PER", "24"))
for M in (4096 * 512, 24576 * 512):
    xyz = torch.rand(M, 3, device=dev) * 1.8 - 0.9
    xyz = (xyz.reshape(-1, 512, 3) * torch.tensor([1.0, 1.0, 0.0], device=dev) + torch.linspace(-0.9, 0.9, 512, device=dev)[None, :, None] * torch.tensor([0.0, 0.0, 1.0], device=dev)).reshape(M, 3)
    jac = None
    if hasattr(lib, "pag_debug_set_jac") or os.environ.get("PAG_LIB_VARIANT"):
        jac = torch.empty(8 * M * per, device=dev, dtype=torch.bfloat16)
        fn = lib.pag_debug_set_jac
        fn.argtypes = [ctypes.c_void_p]
        assert fn(jac.data_ptr()) == 0
    with torch.no_grad():
        for _ in range(3):
            out = ops.encode(xyz, g.tables, g._spec, layout="xcd8", half_coords=g.rounds_coords())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            out = ops.encode(xyz, g.tables, g._spec, layout="xcd8", half_coords=g.rounds_coords())
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
    print("M %9d  forward %.3f ms  (jacobian store: %s)" % (M, dt * 1e3, "%d B per sample" % (8 * per * 2) if jac is not None else "none"))
    if jac is not None:
        go = torch.empty(8, M, 8, device=dev, dtype=torch.bfloat16)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            a = jac.view(torch.int32).sum()
            b = go.view(torch.int32).sum()
        torch.cuda.synchronize()
        print("           streaming read of the jacobian + the gradient (torch sum, an upper bound for a dedicated pass): %.3f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
    del xyz, jac
    torch.cuda.empty_cache()
