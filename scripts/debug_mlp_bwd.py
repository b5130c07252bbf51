import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pagnerf_amd import ops, _lib as L
dev = torch.device("cuda:0")
lib = L.load()
rs = np.random.RandomState(2)
for dims in [(48, 64, 16), (48, 64, 64, 16)]:
    M = 1024
    nl = len(dims) - 1
    W = [torch.from_numpy((rs.standard_normal(size=(dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32)).bfloat16().float() for i in range(nl)]
    b = [torch.from_numpy((rs.standard_normal(size=(dims[i + 1],)) * 0.1).astype(np.float32)) for i in range(nl)]
    x = torch.from_numpy(rs.standard_normal(size=(M, dims[0])).astype(np.float32)).bfloat16().float()
    go = torch.from_numpy(rs.standard_normal(size=(M, dims[-1])).astype(np.float32))
    hs = []; h = x
    for i in range(nl - 1):
        h = torch.relu(h @ W[i].t() + b[i]); hs.append(h)
    y = h @ W[-1].t() + b[-1]
    dz_ref = [None] * nl
    dz_ref[-1] = go
    for i in range(nl - 2, -1, -1):
        dz_ref[i] = (dz_ref[i + 1] @ W[i + 1]) * (hs[i] > 0)
    dx_ref = dz_ref[0] @ W[0]
    # forward on GPU
    xg = x.to(dev); Wg = [w.to(dev).contiguous() for w in W]; bg = [v.to(dev) for v in b]
    out = torch.empty(M, dims[-1], device=dev)
    hid = [torch.empty(M, 64, device=dev, dtype=torch.bfloat16) for _ in range(nl - 1)]
    a = L.MlpFwdArgs(); a.x1, a.x1_dtype, a.k1 = xg.data_ptr(), L.F32, dims[0]
    a.in_dim, a.n_layers, a.out_dim = dims[0], nl, dims[-1]
    for i in range(nl): a.W[i], a.b[i] = Wg[i].data_ptr(), bg[i].data_ptr()
    a.out_act, a.out, a.out_dtype, a.mode = 0, out.data_ptr(), L.F32, L.MLP_MFMA_BF16
    for i, t in enumerate(hid): a.hidden_save[i] = t.data_ptr()
    L.check(lib.pag_mlp_fwd(ctypes.byref(a), M, None), "fwd")
    torch.cuda.synchronize()
    print(dims, "fwd out err", float((out.cpu() - y).abs().max()))
    for i in range(nl - 1):
        print("  hidden", i, "err", float((hid[i].float().cpu() - hs[i]).abs().max()), "mask mismatches", int(((hid[i].float().cpu() > 0) != (hs[i] > 0)).sum()))
    gog = go.to(dev)
    dz = [torch.zeros(M, 64, device=dev, dtype=torch.bfloat16) for _ in range(nl - 1)] + [torch.zeros(M, dims[-1], device=dev, dtype=torch.bfloat16)]
    dx = torch.zeros(M, dims[0], device=dev)
    bb = L.MlpBwdArgs(); bb.grad_out, bb.out, bb.out_dtype, bb.out_act = gog.data_ptr(), out.data_ptr(), L.F32, 0
    bb.k1, bb.in_dim, bb.n_layers, bb.out_dim = dims[0], dims[0], nl, dims[-1]
    for i in range(nl): bb.W[i], bb.dz[i] = Wg[i].data_ptr(), dz[i].data_ptr()
    for i, t in enumerate(hid): bb.hidden_save[i] = t.data_ptr()
    bb.dx1, bb.dx1_dtype, bb.mode = dx.data_ptr(), L.F32, L.MLP_MFMA_BF16
    L.check(lib.pag_mlp_bwd(ctypes.byref(bb), M, None), "bwd")
    torch.cuda.synchronize()
    for i in range(nl):
        got = dz[i].float().cpu(); want = dz_ref[i]
        e = (got - want).abs()
        print("  dz", i, "max err", float(e.max()), "rel", float(e.max() / want.abs().max()), "per-col max err (first 16):", [round(float(v), 3) for v in e.max(0).values[:16]])
        if e.max() > 0.05:
            # find structure: correlation of got columns with want columns
            c = (got.t() @ want) / (got.norm(dim=0)[:, None] * want.norm(dim=0)[None, :] + 1e-9)
            print("    best matching want-col for each got-col:", c.abs().argmax(1).tolist()[:32])
    e = (dx.cpu() - dx_ref).abs()
    print("  dx max err", float(e.max()), "rel", float(e.max() / dx_ref.abs().max()))
