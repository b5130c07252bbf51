import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from pagnerf_amd import ops, _lib as L
from oracle import render as orr
dev = torch.device("cuda:0")
print("allow_tf32", torch.backends.cuda.matmul.allow_tf32, torch.get_float32_matmul_precision())
a = torch.randn(64, 20000, device=dev); b = torch.randn(20000, 48, device=dev)
ref = (a.double() @ b.double())
print("fp32 mm rel err", float(((a @ b).double() - ref).abs().max() / ref.abs().max()))
ab, bb = a.bfloat16(), b.bfloat16()
refb = ab.double() @ bb.double()
try:
    o = torch.mm(ab, bb, out_dtype=torch.float32)
    print("bf16 mm out_dtype f32 ok, rel err", float((o.double() - refb).abs().max() / refb.abs().max()))
except Exception as e:
    print("mm out_dtype unsupported:", repr(e)[:200])
print("bf16 mm -> bf16 rel err", float(((ab @ bb).double() - refb).abs().max() / refb.abs().max()))

# ---- raymarch diffs
rs = np.random.RandomState(3)
N, S, level = 37, 24, 3
o = torch.from_numpy(rs.uniform(-0.6, 0.6, size=(N, 3)).astype(np.float32))
d = rs.standard_normal(size=(N, 3)).astype(np.float32)
d = torch.from_numpy(d / np.linalg.norm(d, axis=1, keepdims=True))
jit = torch.from_numpy(rs.uniform(0, 1, size=(N, S)).astype(np.float32))
ref = orr.raymarch_ray(o, d, 0.0, 2.0, S, jit, None, level)
got = ops.raymarch_ray(o.to(dev), d.to(dev), 0.0, 2.0, S, jit.to(dev), None, level)
for name, g_, r_ in (("samples", got[2].cpu(), ref[2][:, 0]), ("depths", got[3].cpu(), ref[3][:, 0]), ("deltas", got[4].cpu(), ref[4][:, 0])):
    print(name, "equal", torch.equal(g_, r_), "max abs", float((g_ - r_).abs().max()), "n diff", int((g_ != r_).sum()), "of", g_.numel())
depth = torch.linspace(0, 1.0, S)[None] + jit / S
print("lin+jit/S check vs fdiv", float((depth - (torch.linspace(0, 1.0, S)[None] + jit / float(S))).abs().max()))
dd = depth ** 2
print("pow2 vs mul", torch.equal(dd, depth * depth))
samples_a = torch.addcmul(o[:, None], d[:, None], (dd * 2.0)[..., None])
samples_b = o[:, None] + d[:, None] * (dd * 2.0)[..., None]
print("addcmul == mul+add on CPU:", torch.equal(samples_a, samples_b), float((samples_a - samples_b).abs().max()))

# ---- bf16 MLP NL=3 bwd per-tensor errors
import test_gpu_parity as T
rs = np.random.RandomState(2)
for dims, act, k1 in [((48, 64, 64, 16), 0, None), ((48, 64, 64, 3), 0, None), ((48, 64, 64, 3), 1, None), ((43, 64, 64, 3), 1, 16), ((16, 64, 64, 3), 0, None)]:
    M, R = 1037, 50
    W, b = T._rand_mlp(rs, dims)
    in_dim = dims[0]
    if k1 is None:
        x1 = torch.from_numpy(rs.standard_normal(size=(M, in_dim)).astype(np.float32)); x2 = idx = None; xfull = x1
    else:
        x1 = torch.from_numpy(rs.standard_normal(size=(M, k1)).astype(np.float32))
        x2 = torch.zeros(R, 32); x2[:, :in_dim - k1] = torch.from_numpy(rs.standard_normal(size=(R, in_dim - k1)).astype(np.float32))
        idx = torch.from_numpy(np.sort(rs.randint(0, R, size=M)).astype(np.int32))
        xfull = torch.cat([x1, x2[idx.long(), :in_dim - k1]], -1)
    for mode_name in ("fp32", "bf16"):
        mode = L.MLP_FP32 if mode_name == "fp32" else L.MLP_MFMA_BF16
        Wr = [w.bfloat16().float() for w in W] if mode_name == "bf16" else W
        xr = xfull.bfloat16().float() if mode_name == "bf16" else xfull
        Wt = [w.clone().requires_grad_(True) for w in Wr]; bt = [v.clone().requires_grad_(True) for v in b]; xt = xr.clone().requires_grad_(True)
        ref = T._torch_mlp(xt, Wt, bt, act)
        go = torch.from_numpy(rs.standard_normal(size=ref.shape).astype(np.float32)); ref.backward(go)
        Wg = [w.to(dev).requires_grad_(True) for w in W]; bg = [v.to(dev).requires_grad_(True) for v in b]; x1g = x1.to(dev).requires_grad_(True)
        out = ops.fused_mlp(x1g, Wg, bg, x2=None if x2 is None else x2.to(dev), x2_index=None if idx is None else idx.to(dev), in_dim=in_dim, out_act=act, mode=mode)
        out.backward(go.to(dev))
        errs = {"out": float((out.detach().cpu() - ref.detach()).abs().max())}
        n1 = x1.shape[1]
        for name, got_, want in [("dx", x1g.grad.cpu(), xt.grad[:, :n1])] + [("dW%d" % i, Wg[i].grad.cpu(), Wt[i].grad) for i in range(len(W))] + [("db%d" % i, bg[i].grad.cpu(), bt[i].grad) for i in range(len(W))]:
            errs[name] = float((got_ - want).abs().max()) / (float(want.abs().max()) + 1e-12)
        print(dims, act, k1, mode_name, {k: "%.2e" % v for k, v in errs.items()})
