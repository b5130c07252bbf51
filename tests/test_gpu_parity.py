"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle and the golden
vectors captured from the reference.  Need a real MI355X:  pytest -m gpu

Tolerances:
  integer / index work                      bit-exact
  fp32 encode (hash, permuto)               bit-exact against the oracle (same op order, no FMA)
  fp32 decoders / compositing               rtol 1e-5 (+ tiny atol) - summation order differs
  bf16 MFMA decoders                        2e-2 absolute on post-activation values
"""
import numpy as np
import pytest
import torch

from conftest import golden, table_from_seed

pytestmark = pytest.mark.gpu


def _ops():
    from pagnerf_amd import ops, _lib
    return ops, _lib


# ---------------------------------------------------------------------------------------------- encode
def test_hash_encode_matches_reference_golden_bit_exact(gpu_device):
    ops, L = _ops()
    g = golden("g1_hash.npz")
    for tag in ("a", "b"):
        log2T, res = int(g[f"{tag}_log2T"]), [float(r) for r in g[f"{tag}_res"]]
        tab = torch.from_numpy(table_from_seed(int(g[f"{tag}_seed"]), (len(res), 2 ** log2T, 2), str(g[f"{tag}_kind"])))
        x = torch.from_numpy(g[f"{tag}_x"])
        spec = ops.hash_spec(res, log2T, 2)
        out = ops.encode(x.to(gpu_device), tab.to(gpu_device), spec).cpu().numpy()
        ref = g[f"{tag}_feats"]
        assert out.shape == ref.shape
        assert np.array_equal(out, ref), "max abs diff %g" % np.abs(out - ref).max()
        fm = ops.encode(x.to(gpu_device), tab.to(gpu_device), spec, feature_major=True)
        assert fm.stride() == (1, x.shape[0]) and np.array_equal(fm.cpu().numpy(), ref)


def test_hash_encode_variants_and_backward(gpu_device):
    ops, L = _ops()
    from oracle import hash_encode as oh
    rs = np.random.RandomState(0)
    for (Lv, F, log2T) in ((16, 2, 14), (5, 4, 10), (9, 1, 8), (32, 2, 6)):
        res = oh.level_resolutions(16, 512, Lv)
        x = torch.from_numpy(rs.uniform(-1.1, 1.1, size=(777, 3)).astype(np.float32))
        tab = torch.from_numpy(rs.standard_normal(size=(Lv, 2 ** log2T, F)).astype(np.float32))
        if F == 2:
            ref, _ = oh.hash_encode(x, tab, res, log2T)
        else:   # oracle is written for any F
            ref, _ = oh.hash_encode(x, tab, res, log2T)
        spec = ops.hash_spec(res, log2T, F)
        tg = tab.to(gpu_device).requires_grad_(True)
        scale = torch.from_numpy(rs.uniform(0.5, 1.5, size=(Lv * F,)).astype(np.float32))
        out = ops.encode(x.to(gpu_device), tg, spec)
        assert np.array_equal(out.detach().cpu().numpy(), ref.numpy())
        outs = ops.encode(x.to(gpu_device), tg, spec, feat_scale=scale)
        assert np.array_equal(outs.detach().cpu().numpy(), (ref * scale).numpy())
        go = torch.from_numpy(rs.standard_normal(size=ref.shape).astype(np.float32))
        gref = oh.hash_encode_bwd(x, go * scale, 2 ** log2T, res, log2T)
        for algo in ("binned", "atomic"):
            ops.BWD_ALGO, tg.grad = algo, None
            ops.encode(x.to(gpu_device), tg, spec, feat_scale=scale).backward(go.to(gpu_device))
            np.testing.assert_allclose(tg.grad.cpu().numpy(), gref.numpy(), rtol=2e-4, atol=2e-5, err_msg=algo)
        ops.BWD_ALGO = "binned"
        # bf16 output / fp16 tables stay within their rounding
        ob = ops.encode(x.to(gpu_device), tg.detach(), spec, out_dtype=torch.bfloat16).float().cpu()
        np.testing.assert_allclose(ob.numpy(), ref.numpy(), rtol=1e-2, atol=1e-2)
        oh16 = ops.encode(x.to(gpu_device), tg.detach().half(), spec).cpu()
        ref16, _ = oh.hash_encode(x, tab.half().float(), res, log2T)
        assert np.array_equal(oh16.numpy(), ref16.numpy())
    empty = ops.encode(torch.zeros(0, 3, device=gpu_device), tg.detach(), spec)
    assert empty.shape == (0, Lv * F)


def test_permuto_encode_bit_exact_vs_oracle_and_backward(gpu_device):
    ops, L = _ops()
    from oracle import permuto_encode as op
    rs = np.random.RandomState(1)
    for (Lv, F, cap, fine) in ((24, 2, 2 ** 12, 1e-4), (8, 2, 1000, 1e-2), (3, 4, 2 ** 8, 0.1), (17, 1, 2 ** 10, 1e-3)):
        scales = np.geomspace(1.0, fine, Lv)
        sf = op.scale_factors(scales)
        shifts = (rs.standard_normal(size=(Lv, 3)) * 10).astype(np.float32)
        x = rs.uniform(-1, 1, size=(1501, 3)).astype(np.float32)
        x[:8] = [[0, 0, 0], [1, 1, 1], [-1, -1, -1], [0.5, -0.25, 0.125], [1, -1, 0], [1e-6, 0, 0], [0, 1e-6, 0], [-1e-6, 0, 1]]
        tab = rs.standard_normal(size=(Lv, cap, F)).astype(np.float32)
        ref, idx, bary = op.permuto_encode(x, tab, shifts, sf)
        assert idx.min() >= 0 and idx.max() < cap
        np.testing.assert_allclose(bary.sum(-1), 1.0, atol=2e-5)
        spec = ops.permuto_spec(sf, shifts, cap, F)
        tg = torch.from_numpy(tab).to(gpu_device).requires_grad_(True)
        xg = torch.from_numpy(x).to(gpu_device)
        out = ops.encode(xg, tg, spec)
        got = out.detach().cpu().numpy()
        assert np.array_equal(got, ref), "max abs diff %g at L=%d" % (np.abs(got - ref).max(), Lv)
        go = rs.standard_normal(size=ref.shape).astype(np.float32)
        gref = op.permuto_encode_bwd(x, go, cap, shifts, sf)
        for algo in ("binned", "atomic"):
            ops.BWD_ALGO, tg.grad = algo, None
            ops.encode(xg, tg, spec).backward(torch.from_numpy(go).to(gpu_device))
            np.testing.assert_allclose(tg.grad.cpu().numpy(), gref, rtol=2e-4, atol=2e-5, err_msg=algo)
            # bf16 upstream gradient, feature-major (the production layout)
            tg.grad = None
            ops.encode(xg, tg, spec, out_dtype=torch.bfloat16, feature_major=True).backward(torch.from_numpy(go).to(gpu_device).bfloat16())
            gref16 = op.permuto_encode_bwd(x, torch.from_numpy(go).bfloat16().float().numpy(), cap, shifts, sf)
            np.testing.assert_allclose(tg.grad.cpu().numpy(), gref16, rtol=2e-4, atol=2e-5, err_msg=algo + " bf16")
        ops.BWD_ALGO = "binned"
        fm = ops.encode(xg, tg.detach(), spec, out_dtype=torch.bfloat16, feature_major=True)
        np.testing.assert_allclose(fm.float().cpu().numpy(), ref, rtol=1e-2, atol=1e-2)


def test_xcd8_layout_encode_and_decoder_match_strided(gpu_device):
    """The XCD-grouped bf16 [8,M,8] feature layout is a pure re-arrangement: same values, same decoder outputs / gradients."""
    ops, L = _ops()
    from oracle import permuto_encode as op
    rs = np.random.RandomState(7)
    for (Lv, F) in ((24, 2), (16, 2), (13, 1), (8, 4)):
        cap, M = 2 ** 10, 3000
        sf = op.scale_factors(np.geomspace(1.0, 1e-3, Lv))
        shifts = (rs.standard_normal(size=(Lv, 3)) * 10).astype(np.float32)
        spec = ops.permuto_spec(sf, shifts, cap, F)
        x = torch.from_numpy(rs.uniform(-1, 1, size=(M, 3)).astype(np.float32)).to(gpu_device)
        tab = torch.from_numpy(rs.standard_normal(size=(Lv, cap, F)).astype(np.float32)).to(gpu_device)
        cols = ops.xcd8_columns(Lv, F)
        t1 = tab.clone().requires_grad_(True)
        t2 = tab.clone().requires_grad_(True)
        strided = ops.encode(x, t1, spec, out_dtype=torch.bfloat16)
        grouped = ops.encode(x, t2, spec, layout="xcd8")
        assert grouped.shape == (8, M, 8)
        flat = grouped.permute(1, 0, 2).reshape(M, 64)
        for p, c in enumerate(cols):
            if c >= 0:
                assert torch.equal(flat[:, p], strided[:, c]), (Lv, F, p, c)
            else:
                assert float(flat[:, p].abs().max()) == 0.0
        dims = (Lv * F, 64, 64, 10)
        W, b = _rand_mlp(rs, dims)
        outs = []
        for feats, t, grp in ((strided, t1, None), (grouped, t2, (Lv, F))):
            Wg = [w.to(gpu_device).requires_grad_(True) for w in W]
            bg = [v.to(gpu_device).requires_grad_(True) for v in b]
            if grp is None and (Lv * F) % 8:
                feats = torch.nn.functional.pad(feats, (0, (-Lv * F) % 8))
            y = ops.fused_mlp(feats, Wg, bg, in_dim=Lv * F, out_act=L.ACT_SOFTMAX, x1_grouped=grp)
            (y[:, 0].sum() + (y[:, 3] ** 2).sum()).backward()
            outs.append((y.detach(), [w.grad for w in Wg], [v.grad for v in bg], t.grad))
        (y1, w1, b1, g1), (y2, w2, b2, g2) = outs
        np.testing.assert_allclose(y2.cpu().numpy(), y1.cpu().numpy(), rtol=0, atol=5e-4)   # summation order over k differs
        for a_, b_ in list(zip(w1, w2)) + list(zip(b1, b2)) + [(g1, g2)]:
            assert _rel_l2(b_.cpu(), a_.cpu()) < 2e-3, (Lv, F, _rel_l2(b_.cpu(), a_.cpu()))


def test_head_composite_matches_unfused_path(gpu_device):
    """decoder + softmax + per-ray weighted sum as one autograd node (rank-1 upstream gradient inside the decoder
    backward) vs the two-node path (fused_mlp -> composite_feats): same outputs, same gradients."""
    ops, L = _ops()
    rs = np.random.RandomState(21)
    N, Lv, F, M = 70, 24, 2, 70 * 40 + 13
    counts = rs.multinomial(M, np.ones(N) / N)
    counts[5] = 0
    counts[6] += M - counts.sum()
    ridx = torch.from_numpy(np.repeat(np.arange(N), counts).astype(np.int32)).to(gpu_device)
    M = ridx.shape[0]
    csum = np.cumsum(counts)
    pack_start = torch.from_numpy(np.concatenate([[0], csum]).astype(np.int64)).to(gpu_device)
    ray_of_pack = torch.arange(N, dtype=torch.int32, device=gpu_device)
    w = torch.from_numpy(rs.uniform(0, 0.05, size=M).astype(np.float32)).to(gpu_device)
    alpha = torch.from_numpy(rs.uniform(0.1, 1, size=N).astype(np.float32)).to(gpu_device)
    # wide 3-layer head (best.yaml), narrow head, wide 2-layer head, wide head whose width is not a multiple of 8
    for dims, grp in (((48, 64, 64, 200), (Lv, F)), ((48, 64, 6), (Lv, F)), ((48, 64, 200), (Lv, F)), ((48, 64, 64, 100), (Lv, F))):
        W, b = _rand_mlp(rs, dims)
        x = torch.randn(8, M, 8, device=gpu_device).bfloat16()
        cols = ops.xcd8_columns(Lv, F)
        pad = torch.tensor([c < 0 for c in cols], device=gpu_device)
        x.reshape(8, M, 8).permute(1, 0, 2).reshape(M, 64)[:, pad] = 0          # padding positions hold zeros
        gout = torch.from_numpy(rs.standard_normal(size=(N, dims[-1])).astype(np.float32)).to(gpu_device)
        res = []
        for fused in (True, False):
            Wg = [t.to(gpu_device).requires_grad_(True) for t in W]
            bg = [t.to(gpu_device).requires_grad_(True) for t in b]
            xg = x.clone().requires_grad_(True)
            if fused:
                out = ops.head_composite(xg, Wg, bg, w, alpha, ridx, pack_start, ray_of_pack, N, in_dim=dims[0],
                                         out_act=L.ACT_SOFTMAX, x1_grouped=grp)
            else:
                pr = ops.fused_mlp(xg, Wg, bg, in_dim=dims[0], out_act=L.ACT_SOFTMAX, out_dtype=torch.bfloat16, x1_grouped=grp)
                out = ops.composite_feats(pr, w, alpha, pack_start, ray_of_pack, N)
            (out * gout).sum().backward()
            res.append((out.detach(), xg.grad.float(), [t.grad for t in Wg], [t.grad for t in bg]))
        (o1, x1g, w1, b1), (o2, x2g, w2, b2) = res
        # narrow heads: same kernels on both paths, bit-equal; wide heads: the fused path sums fp32 probabilities rebuilt
        # from the hidden layer (pag_head_composite_fwd), the unfused one their bf16-rounded copy
        if dims[-1] <= 64:
            assert torch.equal(o1, o2)
        else:
            np.testing.assert_allclose(o1.cpu().numpy(), o2.cpu().numpy(), rtol=4e-3, atol=1e-6)
        assert float(o1[5].abs().max()) == 0.0                      # empty ray
        for a_, b_ in [(x1g, x2g)] + list(zip(w1, w2)) + list(zip(b1, b2)):
            assert _rel_l2(a_.cpu(), b_.cpu()) < 1e-2, (dims, _rel_l2(a_.cpu(), b_.cpu()))   # unfused path rounds d_feats to bf16


# ------------------------------------------------------------------------------------------------- MLP
def _rand_mlp(rs, dims):
    W = [torch.from_numpy((rs.standard_normal(size=(dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32)) for i in range(len(dims) - 1)]
    b = [torch.from_numpy((rs.standard_normal(size=(dims[i + 1],)) * 0.1).astype(np.float32)) for i in range(len(dims) - 1)]
    return W, b


def _torch_mlp(x, W, b, act, round_hidden=False):
    """fp32 reference; round_hidden emulates the MFMA path's bf16 hidden activations (straight-through)."""
    h = x
    for i in range(len(W)):
        h = torch.nn.functional.linear(h, W[i], b[i])
        if i + 1 < len(W):
            h = torch.relu(h)
            if round_hidden:
                h = h + (h.bfloat16().float() - h).detach()
    if act == 1:
        h = torch.sigmoid(h)
    elif act == 2:
        h = torch.softmax(h, -1)
    return h


def _rel_l2(got, want):
    return float((got - want).norm() / (want.norm() + 1e-20))


@pytest.mark.parametrize("mode_name", ["fp32", "bf16"])
def test_fused_mlp_forward_backward(gpu_device, mode_name):
    ops, L = _ops()
    mode = L.MLP_FP32 if mode_name == "fp32" else L.MLP_MFMA_BF16
    rs = np.random.RandomState(2)
    cases = [((48, 64, 16), 0, None), ((43, 64, 64, 3), 1, 16), ((48, 64, 6), 2, None), ((48, 64, 64, 200), 2, None),
             ((32, 64, 64, 40), 0, None), ((16, 64, 100), 1, None)]
    for dims, act, k1 in cases:
        M, R = 1000 + 37, 50
        W, b = _rand_mlp(rs, dims)
        in_dim = dims[0]
        if k1 is None:
            x1 = torch.from_numpy(rs.standard_normal(size=(M, in_dim)).astype(np.float32))
            x2 = idx = None
            xfull = x1
        else:
            x1 = torch.from_numpy(rs.standard_normal(size=(M, k1)).astype(np.float32))
            x2 = torch.zeros(R, 32)
            x2[:, :in_dim - k1] = torch.from_numpy(rs.standard_normal(size=(R, in_dim - k1)).astype(np.float32))
            idx = torch.from_numpy(np.sort(rs.randint(0, R, size=M)).astype(np.int32))
            xfull = torch.cat([x1, x2[idx.long(), :in_dim - k1]], -1)
        if mode_name == "bf16":   # the kernel rounds inputs and weights to bf16: compare like with like for the tight check
            Wr = [w.bfloat16().float() for w in W]
            xr = xfull.bfloat16().float()
        else:
            Wr, xr = W, xfull
        Wt = [w.clone().requires_grad_(True) for w in Wr]
        bt = [v.clone().requires_grad_(True) for v in b]
        xt = xr.clone().requires_grad_(True)
        ref = _torch_mlp(xt, Wt, bt, act, round_hidden=(mode_name == "bf16"))
        go = torch.from_numpy(rs.standard_normal(size=ref.shape).astype(np.float32))
        ref.backward(go)
        Wg = [w.to(gpu_device).requires_grad_(True) for w in W]
        bg = [v.to(gpu_device).requires_grad_(True) for v in b]
        x1g = x1.to(gpu_device).requires_grad_(True)
        out = ops.fused_mlp(x1g, Wg, bg, x2=None if x2 is None else x2.to(gpu_device),
                            x2_index=None if idx is None else idx.to(gpu_device), in_dim=in_dim, out_act=act, mode=mode)
        out.backward(go.to(gpu_device))
        o = out.detach().cpu()
        if mode_name == "fp32":
            np.testing.assert_allclose(o.numpy(), ref.detach().numpy(), rtol=1e-5, atol=2e-6, err_msg=str(dims))
            tol = dict(rtol=2e-4, atol=2e-4)
        else:
            assert (o - ref.detach()).abs().max() < 2e-2, (dims, float((o - ref.detach()).abs().max()))
            tol = dict(rtol=5e-2, atol=None)
        n1 = x1.shape[1]
        gx_ref = xt.grad[:, :n1]
        for name, got, want in [("dx", x1g.grad.cpu(), gx_ref)] + \
                [("dW%d" % i, Wg[i].grad.cpu(), Wt[i].grad) for i in range(len(W))] + \
                [("db%d" % i, bg[i].grad.cpu(), bt[i].grad) for i in range(len(W))]:
            # relative L2: a ReLU decision that flips on a last-bit difference moves single entries, not the norm
            err = _rel_l2(got, want)
            lim = 1e-4 if mode_name == "fp32" else 2e-2
            assert err < lim, "%s %s %s: rel L2 err %g" % (mode_name, dims, name, err)


def test_g3_nef_forward_against_reference_golden(gpu_device):
    """PanopticDeltaNeF (HIP) vs the reference's rgb_semantics() outputs (g3_nef.npz)."""
    import pagnerf_amd
    from oracle import hash_encode as oh
    g = golden("g3_nef.npz")
    Lv, log2T = int(g["L"]), int(g["log2T"])
    for precision in ("fp32", "bf16"):
        nef = pagnerf_amd.PanopticDeltaNeF(grid_type="HashGridTorch", feature_dim=2, num_lods=Lv, num_classes=6, num_instances=200,
                                           sem_num_layers=1, sem_softmax=True, inst_num_layers=2, inst_softmax=True,
                                           panoptic_features_type="delta", codebook_bitwidth=log2T, precision=precision)
        res = [int(g["res"][0])] * (Lv - 1) + [int(g["res"][-1])]
        for gi, grid in enumerate((nef.grid, nef.delta_grid)):
            grid.init_from_resolutions(res)
            tab = table_from_seed(int(g["seed_main"]) + gi, (Lv, 2 ** log2T, 2), "normal") * np.float32(0.5)
            grid.tables.data.copy_(torch.from_numpy(tab))
        for short in ("density", "color", "semantics", "inst"):
            dec = getattr(nef, "decoder_" + short)
            lins = list(dec.layers) + [dec.lout]
            for i, lin in enumerate(lins):
                lin.weight.data.copy_(torch.from_numpy(g[f"decoder_{short}_w{i}"]))
                lin.bias.data.copy_(torch.from_numpy(g[f"decoder_{short}_b{i}"]))
        nef = nef.to(gpu_device)
        coords = torch.from_numpy(g["coords"]).to(gpu_device)
        ray_d = torch.from_numpy(g["ray_d"]).to(gpu_device)
        with torch.no_grad():
            out = nef(coords=coords, ray_d=ray_d, pidx=None, lod_idx=None, channels={"density", "rgb", "semantics", "inst_embedding"})
            dens = nef(coords=coords, ray_d=ray_d, channels="density")
        assert out["density"].shape == (256, 1, 1) and out["rgb"].shape == (256, 1, 3)
        assert out["semantics"].shape == (256, 6) and out["inst_embedding"].shape == (256, 200)     # Appendix E.11
        assert torch.equal(dens, out["density"])
        tol = dict(rtol=1e-5, atol=2e-6) if precision == "fp32" else dict(rtol=0, atol=2e-2)
        for ch in ("rgb", "semantics", "inst_embedding"):
            np.testing.assert_allclose(out[ch].float().cpu().numpy(), g[ch], err_msg=f"{precision} {ch}", **tol)
        dref = g["density"]
        dtol = dict(rtol=1e-5, atol=2e-6) if precision == "fp32" else dict(rtol=3e-2, atol=3e-2)
        np.testing.assert_allclose(out["density"].float().cpu().numpy(), dref, err_msg=f"{precision} density", **dtol)


# ------------------------------------------------------------------------------------ march + composite
def test_raymarch_matches_oracle_bit_exact(gpu_device):
    ops, L = _ops()
    from oracle import render as orr
    rs = np.random.RandomState(3)
    for (N, S, level, dense) in ((37, 24, 3, False), (130, 100, 5, False), (64, 512, 7, True), (5, 1, 2, False)):
        o = torch.from_numpy(rs.uniform(-0.6, 0.6, size=(N, 3)).astype(np.float32))
        d = rs.standard_normal(size=(N, 3)).astype(np.float32)
        d = torch.from_numpy(d / np.linalg.norm(d, axis=1, keepdims=True))
        jit = torch.from_numpy(rs.uniform(0, 1, size=(N, S)).astype(np.float32))
        R = 2 ** level
        occ = None if dense else torch.from_numpy(rs.uniform(size=(R, R, R)) > 0.4)
        ref = orr.raymarch_ray(o, d, 0.0, 2.0, S, jit, occ, level)
        bits = None
        if occ is not None:
            from pagnerf_amd.grids import OccupancyBLAS
            blas = OccupancyBLAS(level)
            blas.blas_init(occ.reshape(-1))
            assert torch.equal(blas.occupancy_mask(), occ.reshape(-1))
            bits = blas.blas_bits.to(gpu_device)
        got = ops.raymarch_ray(o.to(gpu_device), d.to(gpu_device), 0.0, 2.0, S, jit.to(gpu_device), bits, level)
        ridx, pidx, samples, depths, deltas, boundary, pack_start, ray_of_pack = [t.cpu() for t in got]
        assert torch.equal(ridx.long(), ref[0]) and torch.equal(pidx.long(), ref[1])
        assert torch.equal(samples, ref[2][:, 0]) and torch.equal(depths, ref[3][:, 0]) and torch.equal(deltas, ref[4][:, 0])
        assert torch.equal(boundary, ref[5])
        # raymarch_ray returns one (possibly empty) pack per ray; the non-empty ones are the kaolin-style packs
        ps, rp = ops.packs_from_boundary(got[0], got[5])
        assert ray_of_pack.tolist() == list(range(N)) and int(pack_start[-1]) == ridx.shape[0]
        nonempty = (pack_start[1:] - pack_start[:-1]) > 0
        assert torch.equal(rp.cpu(), ray_of_pack[nonempty]) and torch.equal(ps.cpu()[:-1], pack_start[:-1][nonempty])


def test_voxel_raymarch_matches_oracle(gpu_device):
    ops, L = _ops()
    from oracle import render as orr
    from pagnerf_amd.grids import OccupancyBLAS
    rs = np.random.RandomState(11)
    for (N, k, level, dense, far) in ((61, 2, 3, False, 3.0), (200, 2, 5, False, 2.0), (33, 4, 4, True, 6.0), (17, 1, 2, False, 0.7)):
        o = torch.from_numpy(rs.uniform(-1.3, 1.3, size=(N, 3)).astype(np.float32))
        d = rs.standard_normal(size=(N, 3)).astype(np.float32)
        d = torch.from_numpy(d / np.linalg.norm(d, axis=1, keepdims=True))
        d[0] = torch.tensor([0.0, 0.0, 1.0])            # axis-aligned rays (zero direction components)
        d[1] = torch.tensor([1.0, 0.0, 0.0])
        R = 2 ** level
        occ = None if dense else torch.from_numpy(rs.uniform(size=(R, R, R)) > 0.5)
        ref = orr.raymarch_voxel(o, d, 0.0, far, k, occ, level)
        bits = None
        if occ is not None:
            blas = OccupancyBLAS(level)
            blas.blas_init(occ.reshape(-1))
            bits = blas.blas_bits.to(gpu_device)
        got = [t.cpu() for t in ops.raymarch_voxel(o.to(gpu_device), d.to(gpu_device), 0.0, far, k, bits, level)]
        assert torch.equal(got[0].long(), ref[0]) and torch.equal(got[1].long(), ref[1]), (N, k, level)
        assert torch.equal(got[3], ref[3][..., 0]) and torch.equal(got[4], ref[4][:, 0]) and torch.equal(got[5], ref[5])
        np.testing.assert_allclose(got[2].numpy(), ref[2].numpy(), rtol=0, atol=2e-7)     # fma emulated in fp64 by the oracle
        assert got[2].shape == (ref[0].shape[0], k, 3)


def test_voxel_raymarch_with_travel_filter_and_coarse_grid(gpu_device):
    """pag_raymarch_voxel_* with max_travel == oracle march followed by the tracer's travel filter (:88-108), bit for bit; the
    LDS coarse occupancy changes nothing; the pack table / per-sample ray ids the kernels emit equal what
    mark_pack_boundaries + repeat_interleave would give."""
    ops, L = _ops()
    from oracle import render as orr
    from pagnerf_amd.grids import OccupancyBLAS
    rs = np.random.RandomState(12)
    for (N, k, level, keep_frac, far, travel) in ((300, 2, 5, 0.3, 3.0, 0.5), (257, 2, 7, 0.1, 3.0, 0.25), (64, 3, 6, 0.5, 2.0, 1e9),
                                                  (50, 2, 5, 0.4, 3.0, 0.0), (40, 1, 4, 0.5, 3.0, 0.3)):
        o = torch.from_numpy(rs.uniform(-1.3, 1.3, size=(N, 3)).astype(np.float32))
        d = rs.standard_normal(size=(N, 3)).astype(np.float32)
        d = torch.from_numpy(d / np.linalg.norm(d, axis=1, keepdims=True))
        d[0] = torch.tensor([0.0, 0.0, 1.0])
        R = 2 ** level
        # blobs, so that coarse cells are a mix of empty / partly / fully occupied
        f = rs.uniform(size=(R // 4, R // 4, R // 4)).repeat(4, 0).repeat(4, 1).repeat(4, 2) * 0.7 + rs.uniform(size=(R, R, R)) * 0.3
        occ = torch.from_numpy(f > np.quantile(f, 1 - keep_frac))
        ridx, pidx, samples, depths, deltas, boundary = orr.raymarch_voxel(o, d, 0.0, far, k, occ, level)
        keep = orr.voxel_travel_filter(ridx, depths, travel) if ridx.numel() else torch.zeros(0, dtype=torch.bool)
        ref = (ridx[keep], pidx[keep], samples[keep], depths[keep][..., 0], deltas.reshape(-1, k)[keep].reshape(-1),
               boundary.reshape(-1, k)[keep].reshape(-1))
        blas = OccupancyBLAS(level)
        blas.blas_init(occ.reshape(-1))
        bits = blas.blas_bits.to(gpu_device)
        coarse = ops.occupancy_coarse(bits, level)
        assert (coarse is not None) == (5 <= level <= 8)
        if coarse is not None:      # the coarse grid is the 4x4x4 OR of the fine one
            want = occ.reshape(R // 4, 4, R // 4, 4, R // 4, 4).any(5).any(3).any(1).reshape(-1)
            cb = ((coarse.long()[:, None] & 0xFFFFFFFF) >> torch.arange(32, device=gpu_device)) & 1
            assert torch.equal(cb.reshape(-1)[:want.numel()].bool().cpu(), want)
        outs = []
        for cg in (None, coarse):
            got = ops.raymarch_voxel(o.to(gpu_device), d.to(gpu_device), 0.0, far, k, bits, level, max_travel=travel,
                                     occupancy_coarse_bits=cg, want_packs=True)
            outs.append([t.cpu() for t in got])
        for a, b in zip(*outs):
            assert torch.equal(a, b)
        got = outs[1]
        assert torch.equal(got[0].long(), ref[0]) and torch.equal(got[1].long(), ref[1]), (N, k, level)
        assert torch.equal(got[3], ref[3]) and torch.equal(got[4], ref[4]) and torch.equal(got[5], ref[5])
        np.testing.assert_allclose(got[2].numpy(), ref[2].numpy(), rtol=0, atol=2e-7)
        pack_start, ray_of_pack, ridx_sample, ridx64 = got[6:]
        per_ray = torch.bincount(ref[0], minlength=N) * k
        assert torch.equal(pack_start, torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(per_ray, 0)]))
        assert torch.equal(ray_of_pack.long(), torch.arange(N)) and torch.equal(ridx64, ref[0])
        assert torch.equal(ridx_sample.long(), ref[0].repeat_interleave(k))
        if travel == 0.0:
            assert got[0].numel() == 0          # `0 < 0` is false: the reference's strict filter drops even the first nugget


def test_g4_voxel_mode_trace_against_reference_golden(gpu_device):
    """The reference tracer's voxel-mode path (travel filter :88-108 + compositing) on its own golden inputs,
    through this build's tracer and HIP compositing."""
    import pagnerf_amd
    g = golden("g4_tracer.npz")
    dev = gpu_device
    t = lambda k: torch.from_numpy(g[k]).to(dev)
    v_ridx, v_depths = t("v_ridx").long(), t("v_depths")
    Mv = v_ridx.shape[0]
    v_samples = torch.zeros(Mv, 2, 3, device=dev)
    captured = {}

    class Grid:
        num_lods, active_lods = 4, [0, 1, 2, 3]

        def raymarch(self, rays, level, num_samples, raymarch_type):
            return v_ridx, v_ridx.int(), v_samples, v_depths, t("v_deltas"), t("v_boundary")

    class Nef:
        grid = Grid()

        def get_supported_channels(self):
            return {"rgb", "density"}

        def __call__(self, coords, ray_d, pidx, lod_idx, channels):
            # the reference filters before querying the nef: recover which nuggets survived from pidx
            keep = captured.setdefault("keep", None)
            m = torch.from_numpy(np.asarray(captured["mask"])).to(dev)
            full = {"density": t("v_density")[m], "rgb": t("v_rgb")[m]}
            captured["n"] = coords.shape[0]
            return {c: full[c] for c in channels}

    from oracle import render as orr
    captured["mask"] = orr.voxel_travel_filter(v_ridx.cpu(), v_depths.cpu(), float(g["v_max_travel"])).numpy()
    tr = pagnerf_amd.PanopticPackedRFTracer(ray_max_travel=float(g["v_max_travel"]), raymarch_type="voxel", num_steps=2, bg_color="white")
    rays = pagnerf_amd.Rays(torch.zeros(16, 3, device=dev), torch.ones(16, 3, device=dev), 0.0, 2.0)
    rb = tr(Nef(), channels={"rgb", "depth"}, rays=rays)
    assert captured["n"] == int(g["v_kept"])
    np.testing.assert_allclose(rb.rgb.cpu().numpy(), g["v_out_rgb"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rb.alpha.cpu().numpy(), g["v_out_alpha"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rb.depth.cpu().numpy(), g["v_out_depth"], rtol=1e-5, atol=1e-6)


def test_voxel_mode_end_to_end_after_prune(gpu_device):
    """prune() -> occupancy bitfield -> voxel-mode trace (trainer.py:338-366 sequence) vs the oracle pipeline."""
    from oracle import permuto_encode as op, decoders as od, render as orr
    nef, tracer, rays, occ, jitter = _make_scene(gpu_device, "fp32", N=128, S=2, cap_log2=12, level=4)
    with torch.no_grad():
        nef.decoder_density.lout.bias[0] = 2.9                      # densities straddle the 2.956 prune threshold
        nef.prune(jitter=torch.rand(16 ** 3, 3, generator=torch.Generator().manual_seed(3)).to(gpu_device))
    mask = nef.grid.occupancy_mask().cpu()
    assert 0 < int(mask.sum()) < mask.numel() and torch.equal(mask, nef.delta_grid.occupancy_mask().cpu())
    tracer.raymarch_type, tracer.num_steps, tracer.ray_max_travel = "voxel", 2, 0.6          # trainer.py:362-366
    rays.dist_max = 3.0
    with torch.no_grad():
        rb = tracer(nef, channels={"rgb", "depth", "semantics"}, rays=rays)
    o, d = rays.origins.cpu(), rays.dirs.cpu()
    ridx, pidx, samples, depths, deltas, boundary = orr.raymarch_voxel(o, d, 0.0, 3.0, 2, mask.reshape(16, 16, 16), 4)
    keep = orr.voxel_travel_filter(ridx, depths, 0.6)
    ridx, samples, depths = ridx[keep], samples[keep], depths[keep]
    deltas = deltas.reshape(-1, 2)[keep].reshape(-1, 1)
    boundary = boundary.reshape(-1, 2)[keep].reshape(-1)
    xyz = samples.reshape(-1, 3).numpy()
    xyz = op.half_round(xyz) if nef.grid.half_coords else xyz         # grids/permuto_grid.py:65,71 (autocast cast to half)

    def enc(grid):
        f, _, _ = op.permuto_encode(xyz, grid.tables.detach().cpu().numpy(), grid.random_shift_per_level.cpu().numpy(),
                                    grid.scale_factors(grid.resolutions).numpy())
        return torch.from_numpy(f)
    params = {}
    for short in ("density", "color", "semantics", "inst"):
        W, b = getattr(nef, "decoder_" + short).weights()
        params[short] = ([w.detach().cpu() for w in W], [v.detach().cpu() for v in b])
    rk = ridx.repeat_interleave(2)
    out = od.nef_forward(enc(nef.grid), enc(nef.delta_grid), d[rk], params, {"rgb", "semantics"}, lod_weights=nef.lod_weights)
    comp = orr.composite(128, rk, boundary, out["density"], deltas, depths=depths.reshape(-1, 1), rgb=out["rgb"],
                         semantics=out["semantics"], bg_color="white")
    for ch in ("rgb", "alpha", "depth", "semantics"):
        np.testing.assert_allclose(getattr(rb, ch).cpu().numpy(), comp[ch].numpy(), rtol=2e-4, atol=2e-5, err_msg=ch)


def test_g4_tracer_composite_against_reference_golden(gpu_device):
    ops, L = _ops()
    g = golden("g4_tracer.npz")
    N = int(g["N"])
    t = lambda k: torch.from_numpy(g[k]).to(gpu_device)
    ps, rp = ops.packs_from_boundary(t("ridx").int(), t("boundary"))
    for bg in ("white", "black"):
        alpha, hit, rgb, depth, w = ops.composite(t("density").reshape(-1), t("rgb").reshape(-1, 3), t("deltas").reshape(-1),
                                                  t("depths").reshape(-1), ps, rp, N, bg_white=(bg == "white"))
        sem = ops.composite_feats(t("semantics"), w, alpha, ps, rp, N)
        inst = ops.composite_feats(t("inst_embedding").bfloat16(), w, alpha, ps, rp, N)
        inst32 = ops.composite_feats(t("inst_embedding"), w, alpha, ps, rp, N)
        tol = dict(rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(alpha.cpu().numpy()[:, None], g[f"{bg}_alpha"], **tol)
        np.testing.assert_allclose(rgb.cpu().numpy(), g[f"{bg}_rgb"], **tol)
        np.testing.assert_allclose(depth.cpu().numpy()[:, None], g[f"{bg}_depth"], **tol)
        np.testing.assert_allclose(sem.cpu().numpy(), g[f"{bg}_semantics"], **tol)
        np.testing.assert_allclose(inst32.cpu().numpy(), g[f"{bg}_inst_embedding"], **tol)
        np.testing.assert_allclose(inst.cpu().numpy(), g[f"{bg}_inst_embedding"], rtol=2e-2, atol=1e-3)
        assert np.array_equal(hit.bool().cpu().numpy(), g[f"{bg}_hit"])
        assert float(w.min()) >= 0 and float(alpha.max()) <= 1 + 1e-6


def test_composite_backward_vs_oracle_autograd(gpu_device):
    ops, L = _ops()
    from oracle import render as orr
    rs = np.random.RandomState(4)
    N, S = 70, 90
    counts = rs.randint(0, S, size=N)
    counts[3] = 0
    counts[10] = 200      # longer than 3 wave chunks
    ridx = torch.from_numpy(np.repeat(np.arange(N), counts)).long()
    M = ridx.shape[0]
    boundary = orr.mark_pack_boundaries(ridx)
    mk = lambda *s: torch.from_numpy(rs.uniform(0, 1, size=s).astype(np.float32))
    sigma = (mk(M) * 20 * (mk(M) > 0.3)).requires_grad_(True)
    rgb = mk(M, 3).requires_grad_(True)
    deltas = mk(M) * 0.02
    depths = torch.from_numpy(np.concatenate([np.sort(rs.uniform(0, 2, size=c)) for c in counts]).astype(np.float32))
    sem = torch.softmax(torch.from_numpy(rs.standard_normal(size=(M, 7)).astype(np.float32)), -1).requires_grad_(True)
    for bg in ("white", "black"):
        ref = orr.composite(N, ridx, boundary, sigma, deltas[:, None], depths=depths, rgb=rgb, semantics=sem, bg_color=bg)
        g_rgb, g_depth, g_alpha, g_sem = mk(N, 3), mk(N, 1), mk(N, 1), mk(N, 7)
        loss = (ref["rgb"] * g_rgb).sum() + (ref["depth"] * g_depth).sum() + (ref["alpha"] * g_alpha).sum()
        grads = torch.autograd.grad(loss, [sigma, rgb], retain_graph=True)
        # panoptic: weights and alpha detached (tracer :148-155)
        w_d = ref["weights"].detach()
        a_d = ref["alpha"].detach()
        sem_ref = torch.zeros(N, 7)
        sem_ref[ridx[boundary]] = (a_d[ridx[boundary]] * orr.sum_reduce(w_d * sem, boundary))
        (g_sem_ref,) = torch.autograd.grad((sem_ref * g_sem).sum(), [sem])
        dev = gpu_device
        sg = sigma.detach().to(dev).requires_grad_(True)
        rg = rgb.detach().to(dev).requires_grad_(True)
        ps, rp = ops.packs_from_boundary(ridx.int().to(dev), boundary.to(dev))
        alpha, hit, orgb, odepth, w = ops.composite(sg, rg, deltas.to(dev), depths.to(dev), ps, rp, N, bg_white=(bg == "white"))
        np.testing.assert_allclose(orgb.detach().cpu().numpy(), ref["rgb"].detach().numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(w.cpu().numpy(), ref["weights"].detach().numpy()[:, 0], rtol=2e-5, atol=1e-7)
        l2 = (orgb * g_rgb.to(dev)).sum() + (odepth[:, None] * g_depth.to(dev)).sum() + (alpha[:, None] * g_alpha.to(dev)).sum()
        l2.backward()
        np.testing.assert_allclose(sg.grad.cpu().numpy(), grads[0].numpy(), rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(rg.grad.cpu().numpy(), grads[1].numpy(), rtol=2e-5, atol=1e-6)
        semg = sem.detach().to(dev).requires_grad_(True)
        so = ops.composite_feats(semg, w, alpha.detach(), ps, rp, N)
        np.testing.assert_allclose(so.detach().cpu().numpy(), sem_ref.detach().numpy(), rtol=1e-5, atol=1e-6)
        (so * g_sem.to(dev)).sum().backward()
        np.testing.assert_allclose(semg.grad.cpu().numpy(), g_sem_ref.numpy(), rtol=1e-5, atol=1e-7)


# ------------------------------------------------------------------------------- end to end + properties
def _make_scene(dev, precision, seed=0, L_perm=24, cap_log2=14, N=256, S=64, level=5, heads=(1, 2)):
    """heads = (sem_num_layers, inst_num_layers): best.yaml's (1, 2) by default."""
    import pagnerf_amd
    torch.manual_seed(seed)
    nef = pagnerf_amd.PanopticDeltaNeF(grid_type="PermutoGrid", feature_dim=2, num_lods=L_perm, num_classes=6, num_instances=200,
                                       sem_num_layers=heads[0], sem_softmax=True, inst_num_layers=heads[1], inst_softmax=True,
                                       panoptic_features_type="delta", capacity_log_2=cap_log2, delta_capacity_log_2=cap_log2,
                                       coarsest_scale=1.0, finest_scale=1e-4, blas_level=level, precision=precision)
    gen = torch.Generator().manual_seed(seed)
    for grid in (nef.grid, nef.delta_grid):
        grid.init_from_scales(random_shift=torch.randn(L_perm, 3, generator=gen) * 10,
                              tables=torch.randn(L_perm, 2 ** cap_log2, 2, generator=gen) * 0.3)
    nef = nef.to(dev)
    tracer = pagnerf_amd.PanopticPackedRFTracer(raymarch_type="ray", num_steps=S, bg_color="white")
    o = (torch.rand(N, 3, generator=gen) - 0.5) * 0.6
    d = torch.nn.functional.normalize(torch.randn(N, 3, generator=gen), dim=-1)
    rays = pagnerf_amd.Rays(o.to(dev), d.to(dev), dist_min=0.0, dist_max=2.0)
    occ = torch.rand(2 ** level, 2 ** level, 2 ** level, generator=gen) > 0.35
    for grid in (nef.grid, nef.delta_grid):
        grid.blas_init(occ.reshape(-1))
    jitter = torch.rand(N, S, generator=gen)
    return nef, tracer, rays, occ, jitter


def _oracle_render(nef, rays, occ, jitter, S, channels):
    """The same scene through the CPU oracle, fp32."""
    from oracle import permuto_encode as op, decoders as od, render as orr
    o, d = rays.origins.cpu(), rays.dirs.cpu()
    ridx, pidx, samples, depths, deltas, boundary = orr.raymarch_ray(o, d, rays.dist_min, rays.dist_max, S, jitter, occ, nef.grid.blas_level)
    xyz = samples[:, 0].numpy()
    xyz = op.half_round(xyz) if nef.grid.half_coords else xyz         # grids/permuto_grid.py:65,71 (autocast cast to half)

    def enc(grid):
        sf = grid.scale_factors(grid.resolutions).numpy()
        f, _, _ = op.permuto_encode(xyz, grid.tables.detach().float().cpu().numpy(), grid.random_shift_per_level.cpu().numpy(), sf)
        return torch.from_numpy(f)
    params = {}
    for short in ("density", "color", "semantics", "inst"):
        W, b = getattr(nef, "decoder_" + short).weights()
        params[short] = ([w.detach().cpu() for w in W], [v.detach().cpu() for v in b])
    out = od.nef_forward(enc(nef.grid), enc(nef.delta_grid), d[ridx], params, channels, lod_weights=nef.lod_weights)
    comp = orr.composite(o.shape[0], ridx, boundary, out["density"], deltas, depths=depths, rgb=out.get("rgb"),
                         semantics=out.get("semantics"), inst=out.get("inst_embedding"), bg_color="white")
    return comp, out, ridx


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_end_to_end_render_vs_oracle(gpu_device, precision):
    nef, tracer, rays, occ, jitter = _make_scene(gpu_device, precision)
    chans = {"rgb", "depth", "semantics", "inst_embedding"}
    with torch.no_grad():
        rb = tracer(nef, channels=chans, rays=rays, jitter=jitter.to(gpu_device), stage="val")
    comp, _, _ = _oracle_render(nef, rays, occ, jitter, 64, chans)
    tol = dict(rtol=2e-4, atol=2e-5) if precision == "fp32" else dict(rtol=0, atol=2e-2)
    for ch in ("rgb", "alpha", "depth", "semantics", "inst_embedding"):
        np.testing.assert_allclose(getattr(rb, ch).float().cpu().numpy(), comp[ch].numpy(), err_msg=f"{precision} {ch}", **tol)
    if precision == "fp32":
        assert torch.equal(rb.hit.cpu(), comp["hit"])
    psnr = -10 * np.log10(np.mean((rb.rgb.cpu().numpy() - comp["rgb"].numpy()) ** 2) + 1e-20)
    assert psnr > (60 if precision == "fp32" else 35), psnr


def test_end_to_end_train_step_gradients_fp32(gpu_device):
    """d loss / d {tables, decoder weights} of rgb L1 (trainer.py:443-446) vs autograd through the oracle."""
    from oracle import permuto_encode as op, decoders as od, render as orr
    nef, tracer, rays, occ, jitter = _make_scene(gpu_device, "fp32", N=96, S=32, cap_log2=10)
    gt = torch.rand(96, 3, generator=torch.Generator().manual_seed(5))
    rb = tracer(nef, channels={"rgb"}, rays=rays, jitter=jitter.to(gpu_device), stage="train")
    loss = 10.0 * torch.abs(rb.rgb - gt.to(gpu_device)).mean()
    loss.backward()
    # oracle: differentiable wrt tables through bary weights (indices fixed), decoders through torch
    o, d = rays.origins.cpu(), rays.dirs.cpu()
    ridx, pidx, samples, depths, deltas, boundary = orr.raymarch_ray(o, d, 0.0, 2.0, 32, jitter, occ, nef.grid.blas_level)
    g = nef.grid
    sf = g.scale_factors(g.resolutions).numpy()
    tab = g.tables.detach().cpu().clone().requires_grad_(True)
    xyz = op.half_round(samples[:, 0].numpy()) if g.half_coords else samples[:, 0].numpy()
    _, idx, bary = op.permuto_encode(xyz, tab.detach().numpy(), g.random_shift_per_level.cpu().numpy(), sf)
    idx_t, bary_t = torch.from_numpy(idx.astype(np.int64)), torch.from_numpy(bary)
    feats = torch.cat([(tab[l][idx_t[l]] * bary_t[l][..., None]).sum(1) for l in range(tab.shape[0])], -1)
    params = {}
    leaves = []
    for short in ("density", "color"):
        W, b = getattr(nef, "decoder_" + short).weights()
        Wc = [w.detach().cpu().clone().requires_grad_(True) for w in W]
        bc = [v.detach().cpu().clone().requires_grad_(True) for v in b]
        params[short] = (Wc, bc)
        leaves += [(f"decoder_{short} W{i}", W[i], Wc[i]) for i in range(len(W))] + [(f"decoder_{short} b{i}", b[i], bc[i]) for i in range(len(b))]
    out = od.nef_forward(feats, None, d[ridx], params, {"rgb"}, lod_weights=nef.lod_weights)
    comp = orr.composite(96, ridx, boundary, out["density"], deltas, rgb=out["rgb"], bg_color="white")
    ref_loss = 10.0 * torch.abs(comp["rgb"] - gt).mean()
    ref_loss.backward()
    assert abs(float(loss) - float(ref_loss)) < 1e-4 * max(1.0, abs(float(ref_loss)))
    for name, p_gpu, p_ref in leaves + [("grid.tables", nef.grid.tables, tab)]:
        got, want = p_gpu.grad.cpu(), p_ref.grad
        scale = float(want.abs().max()) + 1e-12
        assert float((got - want).abs().max()) / scale < 2e-3, name
    assert nef.delta_grid.tables.grad is None      # Appendix E.3: rgb loss never reaches the delta grid


def test_properties_at_full_size(gpu_device):
    """Size-independent properties at BASELINE config-2 size (4096 rays x 512 samples)."""
    import pagnerf_amd
    from pagnerf_amd import ops
    nef, tracer, rays, occ, jitter = _make_scene(gpu_device, "bf16", N=4096, S=512, cap_log2=18, level=7)
    for grid in (nef.grid, nef.delta_grid):
        grid.blas_init(torch.ones(128 ** 3, dtype=torch.bool))
    jit = torch.rand(4096, 512, device=gpu_device)
    with torch.no_grad():
        rb = tracer(nef, channels={"rgb", "depth", "semantics", "inst_embedding"}, rays=rays, jitter=jit)
        assert rb.rgb.shape == (4096, 3) and rb.inst_embedding.shape == (4096, 200)
        a = rb.alpha[:, 0]
        assert float(a.min()) >= 0 and float(a.max()) <= 1 + 1e-5
        # composited probabilities are alpha^2-weighted mixtures: rows sum to alpha^2 (Appendix E.12)
        np.testing.assert_allclose(rb.semantics.sum(-1).cpu().numpy(), (a * a).cpu().numpy(), rtol=2e-2, atol=2e-3)
        np.testing.assert_allclose(rb.inst_embedding.sum(-1).cpu().numpy(), (a * a).cpu().numpy(), rtol=2e-2, atol=2e-3)
        # ray-permutation equivariance (bitwise: no atomics, fixed summation order)
        perm = torch.randperm(4096, device=gpu_device)
        rays_p = pagnerf_amd.Rays(rays.origins[perm], rays.dirs[perm], rays.dist_min, rays.dist_max)
        rb_p = tracer(nef, channels={"rgb", "depth"}, rays=rays_p, jitter=jit[perm])
        assert torch.equal(rb_p.rgb, rb.rgb[perm]) and torch.equal(rb_p.depth, rb.depth[perm])
        # determinism
        rb2 = tracer(nef, channels={"rgb", "depth", "semantics", "inst_embedding"}, rays=rays, jitter=jit)
        assert torch.equal(rb2.inst_embedding, rb.inst_embedding) and torch.equal(rb2.rgb, rb.rgb)
        # encode linearity in the table: enc(a*T1 + T2) == a*enc(T1) + enc(T2) up to fp32 rounding
        x = torch.rand(1 << 20, 3, device=gpu_device) * 2 - 1
        g = nef.grid
        t1, t2 = g.tables.detach(), torch.randn_like(g.tables)
        e = lambda t: ops.encode(x, t, g._spec)
        np.testing.assert_allclose(e(2 * t1 + t2).cpu().numpy(), (2 * e(t1) + e(t2)).cpu().numpy(), rtol=1e-4, atol=1e-5)


def test_encode_position_gradients(gpu_device):
    """d loss / d xyz of both encoders (pose optimisation, ba_pipeline.py:85-92): hash vs autograd through the reference
    (golden g7), permuto vs the oracle's closed form; fp32 [M,C] gradients and the bf16 XCD-grouped layout."""
    from pagnerf_amd import ops
    from oracle import permuto_encode as op
    dev = gpu_device
    g = golden("g7_hash_grad.npz")
    res, log2T = [float(r) for r in g["res"]], int(g["log2T"])
    tab = torch.from_numpy(table_from_seed(int(g["seed"]), (len(res), 2 ** log2T, 2), str(g["kind"]))).to(dev).requires_grad_(True)
    x = torch.from_numpy(g["x"]).to(dev).requires_grad_(True)
    spec = ops.hash_spec(res, log2T, 2)
    out = ops.encode(x, tab, spec)
    assert np.array_equal(out.detach().cpu().numpy(), g["feats"])
    out.backward(torch.from_numpy(g["go"]).to(dev))
    np.testing.assert_allclose(x.grad.cpu().numpy(), g["dx"], rtol=1e-4, atol=2e-3)        # |dx| ~ 1e2..1e3 at res 512
    np.testing.assert_allclose(tab.grad.cpu().numpy(), g["dtables"], rtol=1e-4, atol=1e-4)

    rs = np.random.RandomState(12)
    for M, Lv, cap in ((3000, 24, 4099), (777, 5, 256)):
        F = 2
        sf = op.scale_factors(np.geomspace(1.0, 1e-2, Lv))
        shifts = (rs.standard_normal(size=(Lv, 3)) * 10).astype(np.float32)
        xs = rs.uniform(-1, 1, size=(M, 3)).astype(np.float32)
        tb = rs.standard_normal(size=(Lv, cap, F)).astype(np.float32)
        go = rs.standard_normal(size=(M, Lv * F)).astype(np.float32)
        scale = rs.uniform(0.5, 1.5, size=Lv * F).astype(np.float32)
        ref = op.permuto_encode_bwd_xyz(xs, tb, go * scale[None], shifts, sf)
        spec = ops.permuto_spec(sf, shifts, cap, F)
        x = torch.from_numpy(xs).to(dev).requires_grad_(True)
        t = torch.from_numpy(tb).to(dev)
        ops.encode(x, t, spec, torch.from_numpy(scale)).backward(torch.from_numpy(go).to(dev))
        assert _rel_l2(x.grad.cpu(), torch.from_numpy(ref)) < 1e-5, (M, Lv)
        if ops.xcd8_supported(Lv, F):
            x2 = torch.from_numpy(xs).to(dev).requires_grad_(True)
            o = ops.encode(x2, t, spec, torch.from_numpy(scale), layout="xcd8")
            cols = ops.xcd8_columns(Lv, F)
            gg = torch.zeros(64, M)
            for p, c in enumerate(cols):
                if c >= 0:
                    gg[p] = torch.from_numpy(go[:, c])
            gg = gg.reshape(8, 8, M).permute(0, 2, 1).contiguous().bfloat16()
            o.backward(gg.to(dev))
            gob = np.zeros_like(go)
            for p, c in enumerate(cols):
                if c >= 0:
                    gob[:, c] = gg[p // 8, :, p % 8].float().numpy()
            ref2 = op.permuto_encode_bwd_xyz(xs, tb, gob * scale[None], shifts, sf)
            assert _rel_l2(x2.grad.cpu(), torch.from_numpy(ref2)) < 1e-5, (M, Lv)


def _look_at_views(C, gen):
    """C world->camera matrices around the origin (rotation rows orthonormal)."""
    views = []
    for _ in range(C):
        q = torch.linalg.qr(torch.randn(3, 3, generator=gen))[0]
        if torch.det(q) < 0:
            q[2] = -q[2]
        V = torch.eye(4)
        V[:3, :3] = q
        V[:3, 3] = (torch.rand(3, generator=gen) - 0.5) * 0.4
        views.append(V)
    return torch.stack(views)


@pytest.mark.parametrize("mode", ["ray", "voxel"])
def test_pose_gradients_vs_oracle(gpu_device, mode):
    """BAPipeline (ba_pipeline.py:64-92): d loss / d camera_extrinsics through ray transform -> packed samples -> encoder
    position gradient + view embedding -> decoders -> compositing, vs torch autograd over the CPU oracle chain."""
    import pagnerf_amd
    from oracle import permuto_encode as op, decoders as od, render as orr
    dev = gpu_device
    N, S = 128, 32
    nef, tracer, _, occ, jitter = _make_scene(dev, "fp32", L_perm=8, cap_log2=12, N=N, S=S, level=4)
    for grid in (nef.grid, nef.delta_grid):                      # moderate finest scale: smooth enough for a stable comparison
        grid.finest_scale = 0.05
        grid.init_from_scales(random_shift=grid.random_shift_per_level, tables=grid.tables.detach() * 1.0)
    nef = nef.to(dev)
    gen = torch.Generator().manual_seed(5)
    views = _look_at_views(2, gen)
    pipe = pagnerf_amd.BAPipeline(nef, views, tracer, anchor_frame_idxs=[]).to(dev)
    base_o = torch.zeros(N, 3)
    base_d = torch.nn.functional.normalize(torch.randn(N, 3, generator=gen), dim=-1)
    base = pagnerf_amd.Rays(base_o.to(dev), base_d.to(dev), 0.0, 2.0)
    G = torch.randn(N, 3, generator=gen)
    Gd = torch.randn(N, 1, generator=gen)
    chans = {"rgb", "depth"}
    if mode == "ray":
        rb = pipe(rays=base, cam_ids=[0, 1], channels=chans, jitter=jitter[:N, :S].to(dev), stage="train",
                  raymarch_type="ray", num_steps=S)
    else:
        tracer.ray_max_travel = 10.0
        rb = pipe(rays=base, cam_ids=[0, 1], channels=chans, stage="train", raymarch_type="voxel", num_steps=3)
    loss = (rb.rgb * G.to(dev)).sum() + (rb.depth * Gd.to(dev)).sum()
    loss.backward()
    got = pipe.camera_extrinsics.grad.cpu()

    # ---- oracle chain on the CPU
    prm = pipe.camera_extrinsics.detach().cpu().clone().requires_grad_(True)
    from pagnerf_amd.ba_pipeline import rotation_6d_to_matrix
    R = rotation_6d_to_matrix(prm[:, :6])
    o = torch.matmul(base_o.reshape(2, -1, 3) - prm[:, None, 6:], R).reshape(-1, 3)
    d = torch.matmul(base_d.reshape(2, -1, 3), R)
    d = (d / torch.linalg.norm(d, dim=-1, keepdim=True)).reshape(-1, 3)
    if mode == "ray":
        ridx, pidx, samples, depths, deltas, boundary = orr.raymarch_ray(o, d, 0.0, 2.0, S, jitter[:N, :S], occ, 4)
        rk = ridx
    else:
        with torch.no_grad():
            ridx, pidx, s0, depths, deltas, boundary = orr.raymarch_voxel(o.detach(), d.detach(), 0.0, 2.0, 3, occ, 4)
        samples = torch.addcmul(o[ridx][:, None], d[ridx][:, None], depths)           # wisp: o + d * depth, depth constant
        assert torch.allclose(samples.detach(), s0, atol=1e-6)
        rk = ridx.repeat_interleave(3)
    sf = {id(g): g.scale_factors(g.resolutions).numpy() for g in (nef.grid, nef.delta_grid)}

    class Enc(torch.autograd.Function):
        @staticmethod
        def forward(ctx, xyz, grid):
            ctx.grid, ctx.xyz = grid, xyz.detach().numpy()
            if grid.half_coords:       # grids/permuto_grid.py:65,71; autograd passes the gradient straight through the casts
                ctx.xyz = op.half_round(ctx.xyz)
            f, _, _ = op.permuto_encode(ctx.xyz, grid.tables.detach().cpu().numpy(), grid.random_shift_per_level.cpu().numpy(), sf[id(grid)])
            return torch.from_numpy(f)

        @staticmethod
        def backward(ctx, g):
            grid = ctx.grid
            return torch.from_numpy(op.permuto_encode_bwd_xyz(ctx.xyz, grid.tables.detach().cpu().numpy(), g.numpy(),
                                                              grid.random_shift_per_level.cpu().numpy(), sf[id(grid)])), None
    xyz = samples.reshape(-1, 3)
    params = {}
    for short in ("density", "color", "semantics", "inst"):
        W, b = getattr(nef, "decoder_" + short).weights()
        params[short] = ([w.detach().cpu() for w in W], [v.detach().cpu() for v in b])
    out = od.nef_forward(Enc.apply(xyz, nef.grid), Enc.apply(xyz.detach(), nef.delta_grid), d[rk], params, {"rgb"},
                         lod_weights=nef.lod_weights)
    comp = orr.composite(N, rk, boundary, out["density"], deltas.reshape(-1, 1), depths=depths.reshape(-1, 1), rgb=out["rgb"],
                         bg_color="white")
    np.testing.assert_allclose(rb.rgb.detach().cpu().numpy(), comp["rgb"].detach().numpy(), rtol=2e-4, atol=2e-5)
    ref_loss = (comp["rgb"] * G).sum() + (comp["depth"] * Gd).sum()
    ref_loss.backward()
    assert float(prm.grad.abs().max()) > 1e-3
    assert _rel_l2(got, prm.grad) < 2e-3, (got, prm.grad)


def test_pose_gradients_bf16_path_tracks_fp32(gpu_device):
    """Production bf16 path (XCD-grouped features, MFMA decoders, rank-1 head gradients) delivers the same pose gradient as
    the fp32 parity path up to bf16 rounding."""
    import pagnerf_amd
    dev = gpu_device
    N, S = 256, 32
    grads = {}
    for precision in ("fp32", "bf16"):
        nef, tracer, _, occ, jitter = _make_scene(dev, precision, L_perm=24, cap_log2=12, N=N, S=S, level=4)
        for grid in (nef.grid, nef.delta_grid):
            grid.finest_scale = 0.05
            grid.init_from_scales(random_shift=grid.random_shift_per_level, tables=grid.tables.detach() * 1.0)
        nef = nef.to(dev)
        gen = torch.Generator().manual_seed(5)
        pipe = pagnerf_amd.BAPipeline(nef, _look_at_views(4, gen), tracer, anchor_frame_idxs=[0]).to(dev)
        base = pagnerf_amd.Rays(torch.zeros(N, 3, device=dev),
                                torch.nn.functional.normalize(torch.randn(N, 3, generator=gen), dim=-1).to(dev), 0.0, 2.0)
        G = torch.randn(N, 3, generator=gen).to(dev)
        rb = pipe(rays=base, cam_ids=[0, 1, 2, 3], channels={"rgb", "semantics", "inst_embedding"}, jitter=jitter[:N, :S].to(dev),
                  stage="train", raymarch_type="ray", num_steps=S)
        ((rb.rgb * G).sum() + rb.semantics[:, 0].sum() + rb.inst_embedding[:, 3].sum()).backward()
        grads[precision] = pipe.camera_extrinsics.grad.cpu()
        assert float(grads[precision][0].abs().sum()) == 0.0            # anchor frame: masked (ba_pipeline.py:56-60)
        assert nef.grid.tables.grad is not None and nef.delta_grid.tables.grad is not None
    assert _rel_l2(grads["bf16"], grads["fp32"]) < 0.1, grads


def test_head_composite_pair_equals_two_nodes(gpu_device):
    """semantic + instance heads as ONE autograd node with in-kernel accumulation of the input gradient (dx1_accumulate) vs two
    head_composite() nodes whose input gradients autograd adds."""
    ops, L = _ops()
    rs = np.random.RandomState(23)
    N, Lv, F = 50, 24, 2
    counts = rs.randint(1, 90, size=N)
    counts[7] = 0
    M = int(counts.sum())
    ridx = torch.from_numpy(np.repeat(np.arange(N), counts).astype(np.int32)).to(gpu_device)
    pack_start = torch.from_numpy(np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)).to(gpu_device)
    ray_of_pack = torch.arange(N, dtype=torch.int32, device=gpu_device)
    w = torch.from_numpy(rs.uniform(0, 0.05, size=M).astype(np.float32)).to(gpu_device)
    alpha = torch.from_numpy(rs.uniform(0.1, 1, size=N).astype(np.float32)).to(gpu_device)
    x = torch.randn(8, M, 8, device=gpu_device).bfloat16()
    cols = ops.xcd8_columns(Lv, F)
    pad = torch.tensor([c < 0 for c in cols], device=gpu_device)
    x.permute(1, 0, 2).reshape(M, 64)[:, pad] = 0
    dims_a, dims_b = (48, 64, 64, 200), (48, 64, 6)
    Wa, ba = _rand_mlp(rs, dims_a)
    Wb, bb = _rand_mlp(rs, dims_b)
    ga = torch.from_numpy(rs.standard_normal(size=(N, 200)).astype(np.float32)).to(gpu_device)
    gb = torch.from_numpy(rs.standard_normal(size=(N, 6)).astype(np.float32)).to(gpu_device)
    res = []
    for pair in (True, False):
        P = [[t.to(gpu_device).requires_grad_(True) for t in lst] for lst in (Wa, ba, Wb, bb)]
        xg = x.clone().requires_grad_(True)
        if pair:
            oa, ob = ops.head_composite_pair(xg, ((P[0], P[1], 48), (P[2], P[3], 48)), w, alpha, ridx, pack_start, ray_of_pack, N,
                                             x1_grouped=(Lv, F))
        else:
            oa = ops.head_composite(xg, P[0], P[1], w, alpha, ridx, pack_start, ray_of_pack, N, in_dim=48, out_act=L.ACT_SOFTMAX, x1_grouped=(Lv, F))
            ob = ops.head_composite(xg, P[2], P[3], w, alpha, ridx, pack_start, ray_of_pack, N, in_dim=48, out_act=L.ACT_SOFTMAX, x1_grouped=(Lv, F))
        ((oa * ga).sum() + (ob * gb).sum()).backward()
        res.append((oa.detach(), ob.detach(), xg.grad.float(), [t.grad for lst in P for t in lst]))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert _rel_l2(res[0][2].cpu(), res[1][2].cpu()) < 6e-3          # one bf16 rounding of the sum instead of two + one
    for g1, g2 in zip(res[0][3], res[1][3]):
        # same products, but the pair sums them over a different number of per-workgroup slabs (both heads' weight gradients
        # share one launch): equal to fp32 summation-order noise, not bit for bit
        assert torch.allclose(g1, g2, rtol=1e-4, atol=1e-6 * float(g2.abs().max()))


def test_short_training_run_converges_and_bf16_tracks_fp32(gpu_device):
    """60 Adam steps on a small synthetic scene through the whole path (march -> grids -> decoders -> compositing -> losses ->
    backward through every fused node -> Adam): the loss must fall, and the bf16 production path must follow the fp32 parity
    path (same initial weights, same targets)."""
    dev = gpu_device
    losses = {}
    for precision in ("fp32", "bf16"):
        nef, tracer, rays, occ, jitter = _make_scene(dev, precision, seed=3, L_perm=24, cap_log2=12, N=256, S=48, level=4)
        gen = torch.Generator().manual_seed(11)
        gt_rgb = torch.rand(256, 3, generator=gen).to(dev) * 0.5 + 0.25
        gt_sem = torch.randint(0, 6, (256,), generator=gen).to(dev)
        gt_inst = torch.randint(0, 200, (256,), generator=gen).to(dev)
        idx = torch.arange(256, device=dev)
        groups = [{"params": [p for n, p in nef.named_parameters() if "grid" in n], "lr": 3e-2},
                  {"params": [p for n, p in nef.named_parameters() if "grid" not in n], "lr": 3e-3}]
        opt = torch.optim.Adam(groups, eps=1e-15)
        hist = []
        for step in range(60):
            opt.zero_grad(set_to_none=True)
            rb = tracer(nef, channels={"rgb", "depth", "semantics", "inst_embedding"}, rays=rays, jitter=jitter.to(dev), stage="train")
            loss = ((rb.rgb - gt_rgb).abs().mean() + 0.1 * (-torch.log(rb.semantics[idx, gt_sem] + 1e-6)).mean()
                    + 0.1 * (-torch.log(rb.inst_embedding[idx, gt_inst] + 1e-6)).mean())
            loss.backward()
            opt.step()
            hist.append(float(loss.detach()))
        assert all(np.isfinite(hist)), hist
        losses[precision] = hist
    for precision, hist in losses.items():
        assert hist[-1] < 0.8 * hist[0], (precision, hist[0], hist[-1])
    assert abs(losses["bf16"][0] - losses["fp32"][0]) < 0.02 * losses["fp32"][0]
    assert abs(losses["bf16"][-1] - losses["fp32"][-1]) < 0.1 * losses["fp32"][-1], (losses["bf16"][-1], losses["fp32"][-1])


def test_fused_weight_gradients_match_separate_kernels(gpu_device):
    """pag_mlp_bwd with wgrad_workspace (dW / db formed per wave from transposed LDS tiles inside the backward-data kernel) against
    the separate path (dz tensors + pag_mlp_wgrad_batch) on the three narrow production decoders - XCD8 density, strided colour
    with the per-ray view embedding (+ the density column), rank-1 semantic head alone and next to the wide instance head - and
    against an fp32 torch reference.  Same bf16 operands on both GPU paths: only the fp32 summation order differs."""
    ops, L = _ops()
    dev = gpu_device
    rs = np.random.RandomState(31)

    def run(fn, params, fused):
        ops.WGRAD_FUSED = fused
        try:
            for prm in params:
                prm.grad = None
            fn()
            torch.cuda.synchronize()
            return [prm.grad.clone() for prm in params]
        finally:
            ops.WGRAD_FUSED = True

    for M, N in ((32 * 37 + 5, 9), (5, 2), (4096 * 3, 16), (1000, 300)):      # the last: ~3 samples per ray, tiles span ~10 rays
        ridx = torch.from_numpy(np.sort(rs.randint(0, N, size=M)).astype(np.int32)).to(dev)
        counts = torch.bincount(ridx.long(), minlength=N)
        pack_start = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(counts, 0)])
        ray_of_pack = torch.arange(N, dtype=torch.int32, device=dev)
        x8 = torch.from_numpy(rs.standard_normal(size=(8, M, 8)).astype(np.float32)).to(dev)
        x8[:, :, 6:] = 0.0                                   # L = 24, F = 2: elements 6, 7 of every group are padding
        x8 = x8.bfloat16().requires_grad_(True)
        cols = ops.xcd8_columns(24, 2)
        xs = torch.zeros(M, 48, device=dev)
        flat = x8.detach().float().permute(1, 0, 2).reshape(M, 64)
        for pos, c in enumerate(cols):
            if c >= 0:
                xs[:, c] = flat[:, pos]
        # ---- density: 48 -> 64 -> 16, XCD8 input, bf16 output
        W, b = _rand_mlp(rs, (48, 64, 16))
        Wg = [w.to(dev).requires_grad_(True) for w in W]
        bg = [v.to(dev).requires_grad_(True) for v in b]
        g = torch.from_numpy(rs.standard_normal(size=(M, 16)).astype(np.float32)).to(dev).bfloat16()

        def density():
            ops.fused_mlp(x8, Wg, bg, in_dim=48, out_act=L.ACT_NONE, out_dtype=torch.bfloat16, x1_grouped=(24, 2)).backward(g)
        a = run(density, Wg + bg + [x8], True)
        bsep = run(density, Wg + bg + [x8], False)
        for u, v in zip(a, bsep):
            assert _rel_l2(u.float(), v.float()) < 1e-4
        Wt = [w.bfloat16().float().to(dev).requires_grad_(True) for w in W]
        bt = [v.to(dev).clone().requires_grad_(True) for v in b]
        _torch_mlp(xs, Wt, bt, 0, round_hidden=True).backward(g.float())
        for u, v in zip(a[:4], [t.grad for t in Wt + bt]):
            assert _rel_l2(u, v) < 2e-2
        # ---- colour: (16 + 27) -> 64 -> 64 -> 3 sigmoid, strided bf16 x1 + per-ray x2, f32 output, density column
        W, b = _rand_mlp(rs, (43, 64, 64, 3))
        Wg = [w.to(dev).requires_grad_(True) for w in W]
        bg = [v.to(dev).requires_grad_(True) for v in b]
        x1 = torch.from_numpy(rs.standard_normal(size=(M, 16)).astype(np.float32)).to(dev).bfloat16().requires_grad_(True)
        x2 = torch.zeros(N, 32, device=dev)
        x2[:, :27] = torch.from_numpy(rs.standard_normal(size=(N, 27)).astype(np.float32)).to(dev)
        g_rgb = torch.from_numpy(rs.standard_normal(size=(M, 3)).astype(np.float32)).to(dev)
        g_sig = torch.from_numpy(rs.standard_normal(size=(M,)).astype(np.float32)).to(dev)

        def colour():
            rgb, sigma = ops.colour_and_density(x1, Wg, bg, x2, ridx, 43, out_act=L.ACT_SIGMOID, out_dtype=torch.float32)
            ((rgb * g_rgb).sum() + (sigma * g_sig).sum()).backward()
        a = run(colour, Wg + bg + [x1], True)
        bsep = run(colour, Wg + bg + [x1], False)
        for u, v in zip(a, bsep):
            assert _rel_l2(u.float(), v.float()) < 1e-4
        Wt = [w.bfloat16().float().to(dev).requires_grad_(True) for w in W]
        bt = [v.to(dev).clone().requires_grad_(True) for v in b]
        xfull = torch.cat([x1.detach().float(), x2.bfloat16().float()[ridx.long(), :27]], -1)
        (_torch_mlp(xfull, Wt, bt, 1, round_hidden=True) * g_rgb).sum().backward()
        for u, v in zip(a[:6], [t.grad for t in Wt + bt]):
            assert _rel_l2(u, v) < 2e-2
        # ---- semantic head (48 -> 64 -> 6 softmax) composited with rank-1 gradients, alone and next to the instance head
        Ws, bs_ = _rand_mlp(rs, (48, 64, 6))
        Wi, bi = _rand_mlp(rs, (48, 64, 64, 200))
        Wsg = [w.to(dev).requires_grad_(True) for w in Ws]
        bsg = [v.to(dev).requires_grad_(True) for v in bs_]
        Wig = [w.to(dev).requires_grad_(True) for w in Wi]
        big = [v.to(dev).requires_grad_(True) for v in bi]
        wts = torch.rand(M, device=dev)
        alpha = torch.rand(N, device=dev)
        gs = torch.from_numpy(rs.standard_normal(size=(N, 6)).astype(np.float32)).to(dev)
        gi = torch.from_numpy(rs.standard_normal(size=(N, 200)).astype(np.float32)).to(dev)

        def sem():
            o = ops.head_composite(x8, Wsg, bsg, wts, alpha, ridx, pack_start, ray_of_pack, N, in_dim=48, out_act=L.ACT_SOFTMAX,
                                   out_dtype=torch.bfloat16, x1_grouped=(24, 2))
            (o * gs).sum().backward()

        def pair():
            oi, os_ = ops.head_composite_pair(x8, ((Wig, big, 48), (Wsg, bsg, 48)), wts, alpha, ridx, pack_start, ray_of_pack, N,
                                              out_dtype=torch.bfloat16, x1_grouped=(24, 2))
            ((oi * gi).sum() + (os_ * gs).sum()).backward()
        # the instance head runs as two launches with its hidden gradient as a bf16 tensor in between (the rounding the register
        # path applies too) and sums a tile that spans several rays in windows: a few bf16 ulps move.  In the pair the semantic head rides
        # in the instance head's second launch (pag_mlp_bwd_args.pair) with the same roundings as its own launch would apply.
        for fn, prms, lim in ((sem, Wsg + bsg + [x8], 1e-4), (pair, Wsg + bsg + Wig + big + [x8], 5e-4)):
            a = run(fn, prms, True)
            bsep = run(fn, prms, False)
            for k, (u, v) in enumerate(zip(a, bsep)):
                assert _rel_l2(u.float(), v.float()) < lim, (fn.__name__, k)


@pytest.mark.gpu
def test_straight_line_forward_kernels(gpu_device):
    """The dedicated forward kernels of the panoptic nef's decoder shapes (csrc/mlp.hip: mlp_fwd_fast, mlp_fwd_wide_stats - buffer-descriptor
    addressing, no branch in the tile loop) against a plain fp32 torch evaluation on bf16-rounded operands: ragged last tiles, one tile per wave
    and grid-strided launches (the cap is 1536 workgroups x 4 tiles), two and three layers, every output epilogue.  Tolerance: bf16 outputs
    (2^-8 relative) of O(1) values + the bf16 hidden activations."""
    from pagnerf_amd import ops, _lib as L
    dev = gpu_device
    rs = np.random.RandomState(5)

    def ref(x, Ws, bs, act):
        h = x
        for i, (W, b) in enumerate(zip(Ws, bs)):
            h = h.bfloat16().float() @ W.bfloat16().float().t() + b
            if i < len(Ws) - 1:
                h = torch.relu(h)
        return torch.sigmoid(h) if act == L.ACT_SIGMOID else (torch.softmax(h, -1) if act == L.ACT_SOFTMAX else h)

    def mk(dims):
        W, b = _rand_mlp(rs, dims)
        return [w.to(dev) for w in W], [v.to(dev) for v in b]

    cols = ops.xcd8_columns(24, 2)
    for M, N in ((5, 2), (1000, 7), (32 * 6144 + 37, 900)):
        x1 = torch.from_numpy(rs.standard_normal(size=(M, 16)).astype(np.float32)).to(dev).bfloat16()
        x2 = torch.zeros(N, 32, device=dev)
        x2[:, :27] = torch.from_numpy(rs.standard_normal(size=(N, 27)).astype(np.float32)).to(dev)
        idx = torch.from_numpy(np.sort(rs.randint(0, N, size=M)).astype(np.int32)).to(dev)
        for dims in ((43, 64, 64, 3), (43, 64, 4)):
            Ws, bs = mk(dims)
            rgb, sigma = ops.colour_and_density(x1, Ws, bs, x2, idx, 43)
            xin = torch.cat([x1.float(), x2[idx.long()][:, :27]], 1)
            assert rgb.shape == (M, dims[-1]) and torch.equal(sigma, torch.relu(x1[:, 0].float()))
            assert float((rgb - ref(xin, Ws, bs, L.ACT_SIGMOID)).abs().max()) < 2e-3, (M, dims)
        x8 = torch.from_numpy(rs.standard_normal(size=(8, M, 8)).astype(np.float32)).to(dev)
        x8[:, :, 6:] = 0
        x8 = x8.bfloat16()
        xin = torch.zeros(M, 48, device=dev)
        for pos, c in enumerate(cols):
            if c >= 0:
                xin[:, c] = x8[pos // 8, :, pos % 8].float()
        for dims, act, tol in (((48, 64, 16), L.ACT_NONE, 2e-2), ((48, 64, 64, 32), L.ACT_NONE, 2e-2), ((48, 64, 6), L.ACT_SOFTMAX, 4e-3),
                               ((48, 64, 64, 8), L.ACT_SOFTMAX, 4e-3)):
            Ws, bs = mk(dims)
            out = ops.fused_mlp(x8, Ws, bs, in_dim=48, out_act=act, out_dtype=torch.bfloat16, x1_grouped=(24, 2))
            assert out.shape == (M, dims[-1])
            assert float((out.float() - ref(xin, Ws, bs, act)).abs().max()) < tol, (M, dims)
            # the variant that also writes the hidden activations (a backward that does not recompute them): same outputs, bit for bit
            fused_was = ops.WGRAD_FUSED
            try:
                ops.WGRAD_FUSED = False
                out_s = ops.fused_mlp(x8, [w.clone().requires_grad_(True) for w in Ws], bs, in_dim=48, out_act=act, out_dtype=torch.bfloat16,
                                      x1_grouped=(24, 2))
            finally:
                ops.WGRAD_FUSED = fused_was
            assert out_s.requires_grad and torch.equal(out_s.detach(), out), (M, dims)
        # the wide head: statistics + composite (pag_head_composite_fwd rebuilds the probabilities from them) against the dense evaluation
        Ws, bs = mk((48, 64, 64, 200))
        ridx = idx
        counts = torch.bincount(ridx.long(), minlength=N)
        pack_start = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(counts, 0)])
        wts, alpha = torch.rand(M, device=dev), torch.rand(N, device=dev)
        o = ops.head_composite(x8, Ws, bs, wts, alpha, ridx, pack_start, torch.arange(N, dtype=torch.int32, device=dev), N, in_dim=48,
                               out_act=L.ACT_SOFTMAX, out_dtype=torch.bfloat16, x1_grouped=(24, 2))
        dense = ref(xin, Ws, bs, L.ACT_SOFTMAX) * wts[:, None]
        want = torch.zeros(N, 200, device=dev).index_add_(0, ridx.long(), dense) * alpha[:, None]
        assert float((o.float() - want).abs().max()) < 2e-2 * max(1.0, float(want.abs().max())), M


@pytest.mark.gpu
@pytest.mark.parametrize("option", ["pos_encoding", "position", "separate", "appearance", "sum"])
def test_nef_panoptic_feature_types_and_multiscale_sum(gpu_device, option):
    """The nef options outside best.yaml (pc_nerf/panoptic_delta_nef.py:172-173, :210-234; decoder input widths panoptic_nef.py:78-105):
    what the panoptic heads read - embedded position, raw position, the delta grid alone, the appearance features - and
    multiscale_type 'sum'.  HIP nef against oracle.decoders.nef_forward on the nef's own grid features: fp32 path tight, bf16 path loose."""
    import pagnerf_amd
    from oracle import decoders as od
    dev = gpu_device
    torch.manual_seed(11)
    M = 300
    coords = (torch.rand(M, 1, 3, device=dev) * 1.6 - 0.8)
    ray_d = torch.nn.functional.normalize(torch.randn(M, 3, device=dev), dim=-1)
    kw = dict(grid_type="PermutoGrid", feature_dim=2, num_lods=24, num_classes=6, num_instances=200, sem_num_layers=1, sem_softmax=True,
              inst_num_layers=2, inst_softmax=True, capacity_log_2=12, delta_capacity_log_2=12)
    if option == "sum":
        kw.update(multiscale_type="sum", panoptic_features_type="delta")
    else:
        kw.update(panoptic_features_type=option)
    for precision in ("fp32", "bf16"):
        torch.manual_seed(5)
        nef = pagnerf_amd.PanopticDeltaNeF(precision=precision, **kw)
        for g in [nef.grid] + ([nef.delta_grid] if hasattr(nef, "delta_grid") else []):
            g.init_from_scales()
            g.tables.data.normal_(0, 0.5)
        nef = nef.to(dev)
        want_dim = {"pos_encoding": 27, "position": 3, "sum": 2}.get(option, 48)
        assert nef.decoder_semantics.input_dim == want_dim and nef.decoder_inst.input_dim == want_dim
        assert nef.decoder_density.input_dim == (2 if option == "sum" else 48)
        chans = {"density", "rgb", "semantics", "inst_embedding"}
        with torch.no_grad():
            out = nef(coords=coords, ray_d=ray_d, channels=chans)
            feats = nef.grid.interpolate(coords, None).reshape(M, -1).float().cpu()
            dfe = nef.delta_grid.interpolate(coords, None).reshape(M, -1).float().cpu() if hasattr(nef, "delta_grid") else None
        params = {k: tuple([t.detach().float().cpu() for t in lst] for lst in getattr(nef, "decoder_" + n).weights())
                  for k, n in (("density", "density"), ("color", "color"), ("semantics", "semantics"), ("inst", "inst"))}
        ref = od.nef_forward(feats, dfe, ray_d.cpu(), params, chans, panoptic_features_type="delta" if option == "sum" else option,
                             multiscale_sum_levels=24 if option == "sum" else 0, coords=coords.cpu(), pos_multires=4)
        tol = 2e-5 if precision == "fp32" else 3e-2
        for ch, shape in (("density", (M, 1, 1)), ("rgb", (M, 1, 3)), ("semantics", (M, 6)), ("inst_embedding", (M, 200))):
            got = out[ch].float().cpu().reshape(M, -1)
            assert out[ch].shape == shape, (ch, out[ch].shape)
            assert float((got - ref[ch].reshape(M, -1)).abs().max()) < tol * max(1.0, float(ref[ch].abs().max())), (option, precision, ch)
    # the fused head + compositing path applies only when the heads read grouped grid features
    assert nef.can_fuse_panoptic({"semantics", "inst_embedding"}) == (option in ("separate", "appearance"))
