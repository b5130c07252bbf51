"""Delta-density variant on the HIP kernels vs the reference's own outputs (tests/golden/g8_dd.npz, from
pc_nerf/panoptic_dd_nef.py and tracers/panoptic_dd_packed_rf_tracer.py) and vs torch autograd over the CPU oracle."""
import types

import numpy as np
import pytest
import torch

from conftest import golden, table_from_seed

pytestmark = pytest.mark.gpu

_DECS = (("density", "decoder_density"), ("color", "decoder_color"), ("semantics", "decoder_semantics"),
         ("inst", "decoder_inst"), ("delta_density", "decoder_delta_density"))


def _make_nef(g, precision, dev):
    import pagnerf_amd
    Lv, log2T = int(g["L"]), int(g["log2T"])
    nef = pagnerf_amd.PanopticDDensityNeF(grid_type="HashGridTorch", feature_dim=2, num_lods=Lv, num_classes=6, num_instances=20,
                                          sem_num_layers=1, sem_softmax=True, inst_num_layers=2, inst_softmax=True,
                                          delta_num_layers=1, delta_hidden_dim=64, codebook_bitwidth=log2T, blas_level=3,
                                          precision=precision)
    res = [int(g["res"][0])] * (Lv - 1) + [int(g["res"][-1])]
    for gi, grid in enumerate((nef.grid, nef.delta_grid)):
        grid.init_from_resolutions(res)
        grid.tables.data.copy_(torch.from_numpy(table_from_seed(int(g["seed_main"]) + gi, (Lv, 2 ** log2T, 2), "normal") * np.float32(0.5)))
    for _, name in _DECS:
        dec = getattr(nef, name)
        for i, lin in enumerate(list(dec.layers) + [dec.lout]):
            lin.weight.data.copy_(torch.from_numpy(g[f"{name}_w{i}"]))
            lin.bias.data.copy_(torch.from_numpy(g[f"{name}_b{i}"]))
    return nef.to(dev)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_dd_nef_forward_against_reference_golden(gpu_device, precision):
    g = golden("g8_dd.npz")
    nef = _make_nef(g, precision, gpu_device)
    coords, ray_d = torch.from_numpy(g["coords"]).to(gpu_device), torch.from_numpy(g["ray_d"]).to(gpu_device)
    chans = {"density", "rgb", "delta_density", "panoptic_density", "semantics", "inst_embedding"}
    with torch.no_grad():
        out = nef(coords=coords, ray_d=ray_d, pidx=None, lod_idx=None, channels=chans)
        pd = nef(coords=coords, ray_d=ray_d, channels="panoptic_density")
    assert torch.equal(pd, out["panoptic_density"]) and pd.shape == (256, 1, 1)
    tol = dict(rtol=1e-5, atol=3e-6) if precision == "fp32" else dict(rtol=3e-2, atol=3e-2)
    for ch in sorted(chans):
        got = out[ch].float().cpu().numpy()
        np.testing.assert_allclose(got.reshape(g["nef_" + ch].shape), g["nef_" + ch], err_msg=f"{precision} {ch}", **tol)


def test_dd_tracer_against_reference_golden(gpu_device):
    """trace() on the golden's packed scene: a stand-in nef / grid hand the tracer the recorded per-sample channels."""
    import pagnerf_amd
    dev = gpu_device
    g = golden("g8_dd.npz")
    N, S = int(g["t_N"]), int(g["t_S"])
    t = lambda k: torch.from_numpy(g["t_" + k]).to(dev)
    ridx, boundary = t("ridx"), t("boundary")

    class Grid:
        num_lods, active_lods = 4, [0, 1, 2, 3]

        def raymarch(self, rays, level, num_samples, raymarch_type):
            return ridx, ridx.int(), torch.zeros(ridx.shape[0], 1, 3, device=dev), t("depths"), t("deltas"), boundary

    class Nef:
        grid = Grid()

        def get_supported_channels(self):
            return {"density", "rgb", "semantics", "inst_embedding", "panoptic_density"}

        def __call__(self, coords, ray_d, pidx, lod_idx, channels):
            full = {"density": t("density"), "rgb": t("rgb"), "semantics": t("semantics"), "inst_embedding": t("inst_embedding"),
                    "panoptic_density": t("panoptic_density")}
            return full[channels] if isinstance(channels, str) else {c: full[c] for c in channels}

    rays = types.SimpleNamespace(origins=t("origins"), dirs=t("dirs"))
    for bg in ("white", "black"):
        tr = pagnerf_amd.PanopticDDensityPackedRFTracer(ray_sparcity_reg=0.01, raymarch_type="ray", num_steps=S, bg_color=bg)
        rb = tr(Nef(), channels={"rgb", "depth", "semantics", "inst_embedding"}, rays=rays, stage="train")
        for ch in ("rgb", "alpha", "depth", "semantics", "inst_embedding", "ray_sparcity_loss"):
            np.testing.assert_allclose(getattr(rb, ch).cpu().numpy(), g[f"t_{bg}_{ch}"], rtol=1e-5, atol=1e-6, err_msg=f"{bg} {ch}")
        assert np.array_equal(rb.hit.cpu().numpy(), g[f"t_{bg}_hit"])


def test_composite_features_gradients_vs_oracle(gpu_device):
    """ops.composite_features: gradients to the panoptic density AND the features vs autograd over oracle.composite_dd."""
    from pagnerf_amd import ops
    from oracle import render as orr
    dev = gpu_device
    rs = np.random.RandomState(6)
    N = 50
    counts = rs.randint(0, 70, size=N)
    counts[4], counts[9] = 0, 300
    ridx = torch.from_numpy(np.repeat(np.arange(N), counts)).long()
    M = ridx.shape[0]
    boundary = orr.mark_pack_boundaries(ridx)
    mk = lambda *s: torch.from_numpy(rs.uniform(0, 1, size=s).astype(np.float32))
    sigma = (mk(M) * 15 * (mk(M) > 0.3)).requires_grad_(True)
    deltas = mk(M) * 0.02
    for C in (1, 6, 200):
        feat = torch.softmax(torch.from_numpy(rs.standard_normal(size=(M, C)).astype(np.float32)), -1).requires_grad_(True)
        G = torch.from_numpy(rs.standard_normal(size=(N, C)).astype(np.float32))
        ref = orr.composite_dd(N, ridx, boundary, torch.zeros(M), sigma, deltas[:, None], semantics=feat)
        gs, gf = torch.autograd.grad((ref["semantics"] * G).sum(), [sigma, feat])
        sg = sigma.detach().to(dev).requires_grad_(True)
        fg = feat.detach().to(dev).requires_grad_(True)
        ps, rp = ops.packs_from_boundary(ridx.int().to(dev), boundary.to(dev))
        out, alpha = ops.composite_features(sg, deltas.to(dev), fg, ridx.int().to(dev), ps, rp, N)
        np.testing.assert_allclose(out.detach().cpu().numpy(), ref["semantics"].detach().numpy(), rtol=1e-5, atol=1e-6)
        (out * G.to(dev)).sum().backward()
        np.testing.assert_allclose(fg.grad.cpu().numpy(), gf.numpy(), rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(sg.grad.cpu().numpy(), gs.numpy(), rtol=5e-4, atol=5e-5)


def test_dd_train_step_and_prune_smoke(gpu_device):
    """bf16 production path end to end: trace -> loss -> backward reaches both grids and all five decoders; prune ORs the masks."""
    import pagnerf_amd
    g = golden("g8_dd.npz")
    dev = gpu_device
    nef = _make_nef(g, "bf16", dev)
    with torch.no_grad():
        nef.decoder_density.lout.bias[0] = 3.0
    tracer = pagnerf_amd.PanopticDDensityPackedRFTracer(raymarch_type="ray", num_steps=32, bg_color="white")
    gen = torch.Generator().manual_seed(1)
    N = 64
    rays = pagnerf_amd.Rays(((torch.rand(N, 3, generator=gen) - 0.5) * 0.5).to(dev),
                            torch.nn.functional.normalize(torch.randn(N, 3, generator=gen), dim=-1).to(dev), 0.0, 2.0)
    rb = tracer(nef, channels={"rgb", "depth", "semantics", "inst_embedding"}, rays=rays, stage="train")
    loss = rb.rgb.mean() + rb.semantics[:, 1].mean() + rb.inst_embedding[:, 2].mean()
    loss.backward()
    for name, prm in nef.named_parameters():
        assert prm.grad is not None and torch.isfinite(prm.grad).all(), name
    assert float(nef.decoder_delta_density.lout.weight.grad.abs().sum()) > 0 and float(nef.delta_grid.tables.grad.abs().sum()) > 0
    nef.prune(jitter=torch.rand(8 ** 3, 3, generator=gen).to(dev))
    assert torch.equal(nef.grid.blas_bits, nef.delta_grid.blas_bits)


def test_batch_render_chunks_match_single_pass(gpu_device):
    """trainer.batch_render (pc_nerf/trainer.py:637-649): rays are independent, so chunked == unchunked, bit for bit."""
    import pagnerf_amd
    g = golden("g8_dd.npz")
    dev = gpu_device
    nef = _make_nef(g, "fp32", dev)
    tracer = pagnerf_amd.PanopticDDensityPackedRFTracer(raymarch_type="voxel", num_steps=2, bg_color="black")
    pipe = pagnerf_amd.Pipeline(nef, tracer)
    gen = torch.Generator().manual_seed(2)
    N = 150
    rays = pagnerf_amd.Rays(((torch.rand(N, 3, generator=gen) - 0.5) * 0.5).to(dev),
                            torch.nn.functional.normalize(torch.randn(N, 3, generator=gen), dim=-1).to(dev), 0.0, 2.0)
    chans = ["rgb", "depth", "semantics"]
    with torch.no_grad():
        full = pipe(rays=rays, lod_idx=None, channels=chans)
        parts = pagnerf_amd.batch_render(pipe, rays, channels=chans, render_batch=64)
    for ch in ("rgb", "alpha", "depth", "semantics", "hit"):
        assert torch.equal(getattr(full, ch), getattr(parts, ch)), ch
    assert parts.rgb.shape == (N, 3)


@pytest.mark.gpu
def test_affine_xcd8_matches_dense_evaluation(gpu_device):
    """pag_affine_xcd8_fwd / _bwd_dx + the weight-gradient kernels (ops.affine_xcd8: the activation-free decoder_delta_density of
    pc_nerf/panoptic_dd_nef.py:49-56 as ONE affine map of the bf16 [8,M,8] features) against a dense fp32 evaluation and its autograd."""
    from pagnerf_amd import ops
    dev = gpu_device
    rs = np.random.RandomState(9)
    cols = ops.xcd8_columns(24, 2)
    for M, n_out in ((5, 1), (1000, 1), (70001, 3)):
        x8 = torch.from_numpy(rs.standard_normal(size=(8, M, 8)).astype(np.float32)).to(dev)
        x8[:, :, 6:] = 0
        x8 = x8.bfloat16().requires_grad_(True)
        W = torch.from_numpy((rs.standard_normal(size=(n_out, 48)) / 7).astype(np.float32)).to(dev).requires_grad_(True)
        b = torch.from_numpy(rs.standard_normal(size=(n_out,)).astype(np.float32)).to(dev).requires_grad_(True)
        g = torch.from_numpy(rs.standard_normal(size=(M, n_out)).astype(np.float32)).to(dev)
        out = ops.affine_xcd8(x8, W, b, (24, 2))
        (out * g).sum().backward()
        xin = torch.zeros(M, 48, device=dev)
        for pos, c in enumerate(cols):
            if c >= 0:
                xin[:, c] = x8.detach()[pos // 8, :, pos % 8].float()
        xin.requires_grad_(True)
        W2, b2 = W.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
        ref = xin @ W2.t() + b2
        (ref * g).sum().backward()
        assert float((out - ref).detach().abs().max()) < 1e-4 * max(1.0, float(ref.detach().abs().max()))
        dx = torch.zeros(M, 48, device=dev)
        for pos, c in enumerate(cols):
            if c >= 0:
                dx[:, c] = x8.grad[pos // 8, :, pos % 8].float()
            else:
                assert float(x8.grad[pos // 8, :, pos % 8].float().abs().max()) == 0.0
        assert float((dx - xin.grad).abs().max()) < 1e-2 * max(1.0, float(xin.grad.abs().max()))          # bf16 output
        scale = float(W2.grad.abs().max())
        assert float((W.grad - W2.grad).abs().max()) < 1e-2 * scale, (M, n_out)                                # bf16 upstream gradient in the GEMM
        assert float((b.grad - b2.grad).abs().max()) < 1e-2 * max(1.0, float(b2.grad.abs().max()))
