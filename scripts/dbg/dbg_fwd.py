"""Forward decoders against a plain fp32 torch evaluation (bf16-rounded weights and inputs): run twice, with and without PAG_NO_FAST_FWD."""
import numpy as np, torch, sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, R + '/tests')
from pagnerf_amd import ops, _lib as L
dev = torch.device('cuda:0'); rs = np.random.RandomState(3)
def ref(x, Ws, bs, act):
    h = x
    for i, (W, b) in enumerate(zip(Ws, bs)):
        h = h.bfloat16().float() @ W.bfloat16().float().t() + b
        if i < len(Ws) - 1: h = torch.relu(h)
    if act == L.ACT_SIGMOID: h = torch.sigmoid(h)
    if act == L.ACT_SOFTMAX: h = torch.softmax(h, -1)
    return h
def mk(dims):
    Ws = [torch.from_numpy(rs.standard_normal(size=(dims[i + 1], dims[i])).astype(np.float32) / np.sqrt(dims[i])).to(dev) for i in range(len(dims) - 1)]
    bs = [torch.from_numpy(0.1 * rs.standard_normal(size=(dims[i + 1],)).astype(np.float32)).to(dev) for i in range(len(dims) - 1)]
    return Ws, bs
for M, N in ((1000, 7), (4096 * 8 + 5, 300)):
    # colour-like: x1 [M,16] bf16 + x2 [N,32] f32 (27 used), 3 layers, sigmoid, 3 outputs
    x1 = torch.from_numpy(rs.standard_normal(size=(M, 16)).astype(np.float32)).to(dev).bfloat16()
    x2 = torch.zeros(N, 32, device=dev); x2[:, :27] = torch.from_numpy(rs.standard_normal(size=(N, 27)).astype(np.float32)).to(dev)
    idx = torch.from_numpy(np.sort(rs.randint(0, N, size=M)).astype(np.int32)).to(dev)
    Ws, bs = mk((43, 64, 64, 3))
    out, sig = ops.colour_and_density(x1, Ws, bs, x2, idx, 43)
    xin = torch.cat([x1.float(), x2[idx.long()][:, :27]], 1)
    print('colour', M, float((out - ref(xin, Ws, bs, L.ACT_SIGMOID)).abs().max()), 'sigma', float((sig - torch.relu(x1[:, 0].float())).abs().max()))
    # XCD8 heads
    x8 = torch.from_numpy(rs.standard_normal(size=(8, M, 8)).astype(np.float32)).to(dev); x8[:, :, 6:] = 0; x8 = x8.bfloat16()
    cols = ops.xcd8_columns(24, 2)
    xin = torch.zeros(M, 48, device=dev)
    for pos, c in enumerate(cols):
        if c >= 0: xin[:, c] = x8[pos // 8, :, pos % 8].float()
    for name, dims, act, od in (('density', (48, 64, 16), L.ACT_NONE, torch.bfloat16), ('semantic', (48, 64, 6), L.ACT_SOFTMAX, torch.bfloat16),
                                ('density3', (48, 64, 64, 16), L.ACT_NONE, torch.bfloat16)):
        Ws, bs = mk(dims)
        out = ops.fused_mlp(x8, Ws, bs, in_dim=48, out_act=act, out_dtype=od, x1_grouped=(24, 2))
        print(name, M, float((out.float() - ref(xin, Ws, bs, act)).abs().max()))
M = 1000
x8 = torch.from_numpy(rs.standard_normal(size=(8, M, 8)).astype(np.float32)).to(dev); x8[:, :, 6:] = 0; x8 = x8.bfloat16()
xin = torch.zeros(M, 48, device=dev)
for pos, c in enumerate(ops.xcd8_columns(24, 2)):
    if c >= 0: xin[:, c] = x8[pos // 8, :, pos % 8].float()
Ws, bs = mk((48, 64, 16))
out = ops.fused_mlp(x8, Ws, bs, in_dim=48, out_act=L.ACT_NONE, out_dtype=torch.bfloat16, x1_grouped=(24, 2)).float()
rf = ref(xin, Ws, bs, L.ACT_NONE)
err = (out - rf).abs()
print('rows with error', (err.max(1).values > 0.05).nonzero().flatten().tolist()[:40])
print('cols with error', (err.max(0).values > 0.05).nonzero().flatten().tolist())
print(out[0, :8].tolist()); print(rf[0, :8].tolist())
M, N = 64, 3
x1 = torch.from_numpy(rs.standard_normal(size=(M, 16)).astype(np.float32)).to(dev).bfloat16()
x2 = torch.zeros(N, 32, device=dev); x2[:, :27] = torch.from_numpy(rs.standard_normal(size=(N, 27)).astype(np.float32)).to(dev)
idx = torch.from_numpy(np.sort(rs.randint(0, N, size=M)).astype(np.int32)).to(dev)
Ws, bs = mk((43, 64, 64, 3))
for zero in ('none', 'x1', 'x2'):
    a1 = x1 * 0 if zero == 'x1' else x1
    a2 = x2 * 0 if zero == 'x2' else x2
    out, sig = ops.colour_and_density(a1, Ws, bs, a2, idx, 43)
    xin = torch.cat([a1.float(), a2[idx.long()][:, :27]], 1)
    err = (out - ref(xin, Ws, bs, L.ACT_SIGMOID)).abs()
    print(zero, 'max', float(err.max()), 'rows', (err.max(1).values > 0.01).nonzero().flatten().tolist()[:20], 'cols', (err.max(0).values > 0.01).nonzero().flatten().tolist())
