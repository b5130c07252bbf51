"""Edge cases the reference's code paths hit: empty inputs (grids/permuto_grid.py:68-69), rays without samples
(tracer :140-146,:164-176), single-sample and very long packs, ragged tails that are not multiples of the tile sizes."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_empty_inputs_everywhere(gpu_device):
    import pagnerf_amd
    from pagnerf_amd import ops, _lib as L
    dev = gpu_device
    spec = ops.hash_spec([16.0, 32.0], 8, 2)
    tab = torch.randn(2, 256, 2, device=dev, requires_grad=True)
    out = ops.encode(torch.zeros(0, 3, device=dev), tab, spec)
    assert out.shape == (0, 4)
    out.sum().backward()
    assert float(tab.grad.abs().sum()) == 0.0
    W = [torch.randn(64, 48, device=dev), torch.randn(16, 64, device=dev)]
    b = [torch.zeros(64, device=dev), torch.zeros(16, device=dev)]
    assert ops.fused_mlp(torch.zeros(0, 48, device=dev), W, b).shape == (0, 16)
    # no ray hits the cube: zero packed samples, every output keeps its background
    g = pagnerf_amd.PermutoGridHIP(2, capacity_log_2=8, num_lods=4, finest_scale=0.01, blas_level=3)
    g.init_from_scales()
    g = g.to(dev)
    assert g.interpolate(torch.zeros(0, 1, 3, device=dev)).shape == (0, 1, 8)          # permuto_grid.py:68-69
    nef = pagnerf_amd.PanopticDeltaNeF(grid_type="PermutoGrid", num_lods=24, feature_dim=2, num_classes=6, num_instances=200,
                                       inst_num_layers=2, sem_num_layers=1, sem_softmax=True, inst_softmax=True,
                                       panoptic_features_type="delta", capacity_log_2=8, delta_capacity_log_2=8, blas_level=3)
    nef.grid.init_from_scales()
    nef.delta_grid.init_from_scales()
    nef = nef.to(dev)
    tracer = pagnerf_amd.PanopticPackedRFTracer(raymarch_type="ray", num_steps=16, bg_color="white")
    rays = pagnerf_amd.Rays(torch.full((7, 3), 5.0, device=dev), torch.tensor([[1.0, 0, 0]], device=dev).repeat(7, 1), 0.0, 2.0)
    for mode in ("ray", "voxel"):
        tracer.raymarch_type = mode
        rb = tracer(nef, channels={"rgb", "depth", "semantics", "inst_embedding"}, rays=rays)
        assert torch.equal(rb.rgb, torch.ones(7, 3, device=dev)) and float(rb.alpha.abs().sum()) == 0 and not bool(rb.hit.any())
        assert float(rb.depth.abs().sum()) == 0 and float(rb.inst_embedding.abs().sum()) == 0 and rb.inst_embedding.shape == (7, 200)
    tracer.bg_color = "black"
    tracer.raymarch_type = "ray"
    assert float(tracer(nef, channels={"rgb"}, rays=rays).rgb.abs().sum()) == 0


def test_ragged_packs_single_and_long(gpu_device):
    """packs of 0, 1, 63, 64, 65 and 5000 samples; M not a multiple of 32 / 64 / 1024."""
    from pagnerf_amd import ops
    from oracle import render as orr
    dev = gpu_device
    rs = np.random.RandomState(5)
    counts = np.array([0, 1, 63, 64, 65, 0, 5000, 2, 1, 0, 129, 7])
    N = len(counts)
    ridx = torch.from_numpy(np.repeat(np.arange(N), counts)).long()
    M = ridx.shape[0]
    boundary = orr.mark_pack_boundaries(ridx)
    mk = lambda *s: torch.from_numpy(rs.uniform(0, 1, size=s).astype(np.float32))
    sigma, rgb, deltas, depths = mk(M) * 5, mk(M, 3), mk(M) * 0.01, mk(M)
    feat = torch.softmax(torch.from_numpy(rs.standard_normal(size=(M, 200)).astype(np.float32)), -1)
    ref = orr.composite(N, ridx, boundary, sigma, deltas[:, None], depths=depths, rgb=rgb, inst=feat)
    offs = np.concatenate([[0], np.cumsum(counts)])
    pack_start = torch.from_numpy(offs.astype(np.int64)).to(dev)
    ray_of_pack = torch.arange(N, dtype=torch.int32, device=dev)
    alpha, hit, orgb, odepth, w = ops.composite(sigma.to(dev), rgb.to(dev), deltas.to(dev), depths.to(dev), pack_start, ray_of_pack, N)
    np.testing.assert_allclose(orgb.cpu().numpy(), ref["rgb"].numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(alpha.cpu().numpy(), ref["alpha"].numpy()[:, 0], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(odepth.cpu().numpy(), ref["depth"].numpy()[:, 0], rtol=1e-5, atol=1e-6)
    assert torch.equal(hit.bool().cpu(), ref["hit"])
    for dt, tol in ((torch.float32, 1e-5), (torch.bfloat16, 2e-2)):
        out = ops.composite_feats(feat.to(dev).to(dt), w, alpha, pack_start, ray_of_pack, N)
        np.testing.assert_allclose(out.cpu().numpy(), ref["inst_embedding"].numpy(), rtol=tol, atol=tol * 1e-1)
    # kaolin-style packs (non-empty only) give the same buffers
    ps2, rp2 = ops.packs_from_boundary(ridx.int().to(dev), boundary.to(dev))
    a2, _, rgb2, _, _ = ops.composite(sigma.to(dev), rgb.to(dev), deltas.to(dev), depths.to(dev), ps2, rp2, N)
    assert torch.equal(a2, alpha) and torch.equal(rgb2, orgb)


def test_permuto_fp16_tables_and_odd_sizes(gpu_device):
    from pagnerf_amd import ops
    from oracle import permuto_encode as op
    dev = gpu_device
    rs = np.random.RandomState(8)
    for M in (1, 31, 33, 1023, 1025, 2049):
        Lv, F, cap = 24, 2, 4099          # prime capacity: modulo path, ragged last slice
        sf = op.scale_factors(np.geomspace(1.0, 1e-4, Lv))
        shifts = (rs.standard_normal(size=(Lv, 3)) * 10).astype(np.float32)
        x = rs.uniform(-1, 1, size=(M, 3)).astype(np.float32)
        tab = rs.standard_normal(size=(Lv, cap, F)).astype(np.float32)
        spec = ops.permuto_spec(sf, shifts, cap, F)
        t16 = torch.from_numpy(tab).half().to(dev).requires_grad_(True)
        ref, _, _ = op.permuto_encode(x, t16.detach().float().cpu().numpy(), shifts, sf)
        out = ops.encode(torch.from_numpy(x).to(dev), t16, spec)
        assert np.array_equal(out.detach().cpu().numpy(), ref), M
        go = rs.standard_normal(size=ref.shape).astype(np.float32)
        out.backward(torch.from_numpy(go).to(dev))
        gref = op.permuto_encode_bwd(x, go, cap, shifts, sf)
        np.testing.assert_allclose(t16.grad.float().cpu().numpy(), gref, rtol=2e-3, atol=2e-3)     # fp16 gradient storage
        xc = ops.encode(torch.from_numpy(x).to(dev), t16.detach(), spec, layout="xcd8")
        assert xc.shape == (8, M, 8)


def test_occupancy_update_bit_exact(gpu_device):
    """pag_occupancy_update vs the torch sequence of panoptic_delta_nef.py:74-75,90-104 (two prunes: EMA-max carries over)."""
    from pagnerf_amd import ops
    dev = gpu_device
    rs = np.random.RandomState(2)
    for cells in (8, 64, 4096, 16 ** 3 + 0, 100000):          # 8 = blas_level 1 (one partial word)
        occ_ref = torch.zeros(cells)
        occ = torch.zeros(cells, device=dev)
        bits = torch.empty(max(1, (cells + 31) // 32), dtype=torch.int32, device=dev)
        for it in range(2):
            dens = torch.from_numpy((rs.standard_normal(size=(cells, 1, 1)) * 3 + 2).astype(np.float32))
            occ_ref = torch.stack([dens[:, 0, 0], occ_ref * 0.6], -1).max(dim=-1)[0]
            mask = occ_ref > (0.01 * 512) / np.sqrt(3)
            ops.occupancy_update(dens.to(dev), occ, bits, 0.6, (0.01 * 512) / np.sqrt(3))
            assert torch.equal(occ.cpu(), occ_ref)
            got = ((bits.cpu().long()[:, None] >> torch.arange(32)) & 1).bool().reshape(-1)[:cells]
            assert torch.equal(got, mask), (cells, it)


def test_encode_with_addend_equals_separate_bf16_add(gpu_device):
    """pag_*_encode_fwd_add: bf16(addend + bf16(features)) in one launch == the tensor add of panoptic_delta_nef.py:226."""
    from pagnerf_amd import ops
    from oracle import permuto_encode as op
    dev = gpu_device
    rs = np.random.RandomState(31)
    M = 5000
    x = torch.from_numpy(rs.uniform(-1, 1, size=(M, 3)).astype(np.float32)).to(dev)
    addend = torch.randn(8, M, 8, device=dev).bfloat16()
    sf = op.scale_factors(np.geomspace(1.0, 1e-3, 24))
    shifts = (rs.standard_normal(size=(24, 3)) * 10).astype(np.float32)
    pspec = ops.permuto_spec(sf, shifts, 1 << 12, 2)
    hspec = ops.hash_spec([16.0 * 1.4 ** i for i in range(16)], 12, 2)
    for spec, Lv in ((pspec, 24), (hspec, 16)):
        tab = torch.randn(Lv, 1 << 12, 2, device=dev).requires_grad_(True)
        plain = ops.encode(x, tab, spec, layout="xcd8")
        fused = ops.encode(x, tab, spec, layout="xcd8", addend=addend)
        assert torch.equal(fused, addend + plain)
        g = torch.randn_like(fused)
        (g1,) = torch.autograd.grad(fused, tab, g)
        (g2,) = torch.autograd.grad(addend + plain, tab, g)
        assert torch.equal(g1, g2)


def test_pack_offsets_and_view_embed_kernels(gpu_device):
    """pag_pack_offsets against torch.cumsum (integers: exact) and pag_view_embed against the tensor-op form of wisp's
    PositionalEmbedder on -ray_d (sin / cos of the same fp32 argument: 1e-6 absolute, the argument scaling is exact)."""
    import ctypes
    from pagnerf_amd import ops, _lib as L
    from pagnerf_amd.nef import positional_embed
    dev = gpu_device
    g = torch.Generator().manual_seed(11)
    for N in (0, 1, 5, 1023, 1024, 1025, 4096, 100003):
        counts = torch.randint(0, 600, (N,), generator=g, dtype=torch.int32).to(dev)
        out = torch.empty(N + 1, device=dev, dtype=torch.int64)
        L.check(L.load().pag_pack_offsets(L.ptr(counts) if N else None, N, L.ptr(out), None, L.stream()), "pag_pack_offsets")
        ref = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(counts.cpu().long(), 0)])
        assert torch.equal(out.cpu(), ref), N
    for R, nf in ((0, 4), (1, 4), (4096, 4), (777, 0), (513, 10)):
        d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1).to(dev)
        width = 3 + 6 * nf
        width += (-width) % 8
        got = ops.view_embed(d, nf, width)
        ref = positional_embed(-d, nf) if nf else -d
        assert got.shape == (R, width)
        assert torch.allclose(got[:, :ref.shape[1]], ref, rtol=0.0, atol=1e-6)
        assert bool((got[:, ref.shape[1]:] == 0).all())


def test_table_gradient_overwrite_mode_writes_every_row(gpu_device):
    """The binned encode backward fills an UNINITIALISED gradient table (pag_*_encode_bwd_set): rows no sample touches, and
    whole levels that receive no gradient, must come back as exact zeros.  The caching allocator is poisoned with NaNs first so
    that an unwritten row cannot look right by accident; the touched rows are checked against the accumulating entry point."""
    from pagnerf_amd import ops, grids
    dev = gpu_device
    g = torch.Generator().manual_seed(2)
    for kind in ("permuto", "hash"):
        L_, F_, M = 16, 2, 300
        if kind == "permuto":
            cap = 1 << 14
            spec = ops.permuto_spec(grids.PermutoGridHIP.scale_factors(np.geomspace(1.0, 1e-3, L_)), torch.randn(L_, 3, generator=g) * 10, cap, F_)
        else:
            cap = 1 << 14
            spec = ops.hash_spec(grids.HashGridHIP.level_resolutions(16, 512, L_), 14, F_)
        tab = (torch.randn(L_, cap, F_, generator=g) * 1e-2).to(dev).requires_grad_(True)
        xyz = ((torch.rand(M, 3, generator=g) - 0.5) * 1.6).to(dev)
        go = torch.randn(M, L_ * F_, generator=g)
        go[:, 4 * F_:6 * F_] = 0                                     # two levels without any gradient
        go = go.to(dev)
        # reference: accumulating entry point into a zeroed table
        ref = torch.zeros(L_, cap, F_, device=dev)
        ops._encode_bwd(spec, xyz, go, None, ref)
        for _ in range(3):
            poison = torch.full((L_ * cap * F_ + 1024,), float("nan"), device=dev)
            del poison
            out = ops.encode(xyz, tab, spec, None, torch.float32)
            (gt,) = torch.autograd.grad(out, tab, go)
            assert torch.isfinite(gt).all()
            assert torch.equal(gt, ref)
            assert float(gt[4:6].abs().max()) == 0.0
            touched = (gt != 0).any(-1).float().mean().item()
            assert 0 < touched < 0.2


def test_rays_without_samples_keep_the_background_without_prefilled_buffers(gpu_device):
    """'ray' mode hands the compositing kernels one pack per ray (empty packs included); they write every ray themselves, so
    the output buffers are no longer pre-filled.  Rays that miss the volume must still come back as background / zeros
     - checked with a NaN-poisoned allocator so that an unwritten element cannot pass by accident."""
    import pagnerf_amd
    from test_gpu_parity import _make_scene
    dev = gpu_device
    nef, tracer, rays, occ, jitter = _make_scene(dev, "bf16", N=256, S=48, cap_log2=12)
    o, d = rays.origins.clone(), rays.dirs.clone()
    o[::2] = torch.tensor([5.0, 5.0, 5.0], device=dev)               # every other ray starts far outside and points away
    d[::2] = torch.nn.functional.normalize(torch.tensor([1.0, 0.5, 0.25], device=dev), dim=0)
    rays2 = pagnerf_amd.Rays(o, d, dist_min=0.0, dist_max=2.0)
    for bg in ("white", "black"):
        tracer.bg_color = bg
        for _ in range(2):
            poison = torch.full((1 << 22,), float("nan"), device=dev)
            del poison
            with torch.no_grad():
                rb = tracer(nef, channels={"rgb", "depth", "semantics", "inst_embedding"}, rays=rays2, jitter=jitter.to(dev), stage="val")
            miss = ~rb.hit
            assert bool(miss[::2].all()) and bool(rb.hit[1::2].any())
            for ch in ("rgb", "depth", "alpha", "semantics", "inst_embedding"):
                assert bool(torch.isfinite(getattr(rb, ch)).all()), ch
            assert torch.equal(rb.rgb[miss], torch.full_like(rb.rgb[miss], 1.0 if bg == "white" else 0.0))
            for ch in ("depth", "alpha", "semantics", "inst_embedding"):
                assert float(getattr(rb, ch)[miss].abs().max()) == 0.0, ch


def test_pose_rays_kernels_vs_tensor_ops(gpu_device):
    """pag_pose_rays_fwd / _bwd (pc_nerf/ba_pipeline.py:85-92 as one launch each way) against the tensor-op restatement of the same map and torch
    autograd through it: per-ray camera indices in arbitrary order, per-image indices (rays_per_entry = rays per image), cameras without a ray
    (zero gradient row), repeated cameras, non-unit a1 / non-orthogonal a2 (the Gram-Schmidt chain), bitwise reproducible gradients."""
    from pagnerf_amd import ops
    from pagnerf_amd.ba_pipeline import BAPipeline, rotation_6d_to_matrix
    dev = gpu_device
    gen = torch.Generator().manual_seed(11)
    C, N = 7, 1000
    prm = torch.randn(C, 9, generator=gen)
    prm[:, :3] *= 1.7                                                     # |a1| != 1
    oc = torch.randn(N, 3, generator=gen) * 0.1
    dc = torch.randn(N, 3, generator=gen)
    cam = torch.randint(0, C - 2, (N,), generator=gen)                    # cameras C-2, C-1 see no ray
    go, gd = torch.randn(N, 3, generator=gen), torch.randn(N, 3, generator=gen)

    def ref(p, cam_idx):
        R = rotation_6d_to_matrix(p[:, :6]).index_select(0, cam_idx)
        t = p[:, 6:].index_select(0, cam_idx)
        v = oc.double() - t
        o = v[:, 0:1] * R[:, 0] + v[:, 1:2] * R[:, 1] + v[:, 2:3] * R[:, 2]
        d = dc.double()[:, 0:1] * R[:, 0] + dc.double()[:, 1:2] * R[:, 1] + dc.double()[:, 2:3] * R[:, 2]
        return o, d / torch.linalg.norm(d, dim=-1, keepdim=True)
    for rpe, cam_t in ((1, cam), (125, torch.tensor([0, 3, 3, 1, 4, 0, 2, 4]))):
        cam_ray = cam_t if rpe == 1 else cam_t.repeat_interleave(rpe)
        pr = prm.double().clone().requires_grad_(True)
        o_r, d_r = ref(pr, cam_ray)
        ((o_r * go.double()).sum() + (d_r * gd.double()).sum()).backward()
        pg = prm.to(dev).requires_grad_(True)
        o_g, d_g = ops.pose_rays(pg, cam_t.int().to(dev), rpe, oc.to(dev), dc.to(dev))
        np.testing.assert_allclose(o_g.detach().cpu().numpy(), o_r.detach().float().numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(d_g.detach().cpu().numpy(), d_r.detach().float().numpy(), rtol=1e-5, atol=1e-6)
        ((o_g * go.to(dev)).sum() + (d_g * gd.to(dev)).sum()).backward()
        want = pr.grad.float()
        np.testing.assert_allclose(pg.grad.cpu().numpy(), want.numpy(), rtol=2e-4, atol=2e-4 * float(want.abs().max()))
        assert float(pg.grad[C - 2:].abs().max()) == 0.0                 # cameras without a ray: rows written, zero
        first = pg.grad.clone()
        pg.grad = None
        o2, d2 = ops.pose_rays(pg, cam_t.int().to(dev), rpe, oc.to(dev), dc.to(dev))
        ((o2 * go.to(dev)).sum() + (d2 * gd.to(dev)).sum()).backward()
        assert torch.equal(pg.grad, first)                                # fixed summation order
        # only one of the two outputs used: the other gradient is None, not a zero tensor
        pg.grad = None
        o3, _ = ops.pose_rays(pg, cam_t.int().to(dev), rpe, oc.to(dev), dc.to(dev))
        (o3 * go.to(dev)).sum().backward()
        pr2 = prm.double().clone().requires_grad_(True)
        (ref(pr2, cam_ray)[0] * go.double()).sum().backward()
        np.testing.assert_allclose(pg.grad.cpu().numpy(), pr2.grad.float().numpy(), rtol=2e-4, atol=2e-4 * float(pr2.grad.abs().max()))
    # BAPipeline on the GPU uses the kernels in both forms and they agree with each other
    views = torch.eye(4).repeat(4, 1, 1)
    views[:, :3, 3] = torch.randn(4, 3, generator=gen) * 0.1
    pipe = BAPipeline(torch.nn.Module(), views, anchor_frame_idxs=[0]).to(dev)
    import pagnerf_amd
    base = pagnerf_amd.Rays(oc[:800].to(dev), dc[:800].to(dev))
    a = pipe.transform_rays(base, [2, 0, 3, 1])
    b = pipe.transform_rays_indexed(oc[:800].to(dev), dc[:800].to(dev), torch.tensor([2, 0, 3, 1]).repeat_interleave(200).to(dev))
    assert torch.equal(a.origins, b.origins) and torch.equal(a.dirs, b.dirs) and a.origins.requires_grad
    (a.origins.sum() + a.dirs[:, 0].sum()).backward()
    assert float(pipe.camera_extrinsics.grad[0].abs().max()) == 0.0 and float(pipe.camera_extrinsics.grad[1:].abs().min()) >= 0.0     # anchor frame masked (:56-60)


def test_view_embed_gradient_vs_tensor_ops(gpu_device):
    """ops.view_embed_grad: forward = pag_view_embed (the gradient-free path's values), backward = pag_view_embed_bwd, against torch autograd through the
    tensor-op form of wisp's PositionalEmbedder on -dirs."""
    from pagnerf_amd import ops
    from pagnerf_amd.nef import positional_embed
    dev = gpu_device
    gen = torch.Generator().manual_seed(2)
    for n_freq, R in ((4, 1000), (0, 5), (6, 33)):
        d = torch.nn.functional.normalize(torch.randn(R, 3, generator=gen), dim=-1)
        width = 3 + 6 * n_freq
        width += (-width) % 8
        g = torch.randn(R, width, generator=gen)
        dr = d.double().clone().requires_grad_(True)
        pe = positional_embed(-dr, n_freq) if n_freq else -dr
        (pe * g[:, :pe.shape[1]].double()).sum().backward()
        dg = d.to(dev).requires_grad_(True)
        out = ops.view_embed_grad(dg, n_freq, width)
        assert torch.equal(out.detach(), ops.view_embed(d.to(dev), n_freq, width))
        (out * g.to(dev)).sum().backward()
        np.testing.assert_allclose(dg.grad.cpu().numpy(), dr.grad.float().numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("pose", [False, True])
def test_zero_gradient_tiles_take_the_early_outs_with_the_same_gradients(gpu_device, pose):
    """Samples whose upstream gradient is exactly zero (in a trained scene: empty space, sigma = relu(pre) = 0; here: rays the loss does not use) make the
    backward kernels skip whole waves of the table gradient's bin pass and the row requests of the position gradient.
    The skipped work contributes exact zeros, so: gradients of `loss over the rays in S` computed on the FULL batch (most tiles dead: runs of 24 unused rays =
    24 x 64 samples) equal - up to the fp32 summation order of the weight-gradient slabs - the gradients of the same loss on a batch that holds only the
    rays of S (nothing to skip), for every parameter and, with pose optimisation, for the camera extrinsics."""
    import pagnerf_amd
    import test_gpu_parity as T
    from pagnerf_amd.ba_pipeline import BAPipeline
    dev = gpu_device
    N, S = 256, 64
    nef, tracer, rays, occ, jitter = T._make_scene(dev, "bf16", N=N, S=S, cap_log2=12)
    keep = (torch.arange(N) % 32) >= 24                       # 8 of every 32 rays carry the loss: runs of 24 x 64 = 1536 samples without a gradient
    gen = torch.Generator().manual_seed(4)
    G = torch.randn(N, 3, generator=gen).to(dev) * keep[:, None].to(dev)
    Gd = torch.randn(N, 1, generator=gen).to(dev) * keep[:, None].to(dev)
    views = torch.eye(4).repeat(2, 1, 1)
    views[:, :3, 3] = torch.tensor([[0.01, -0.02, 0.0], [0.0, 0.015, -0.01]])
    pipe = BAPipeline(nef, views, tracer=tracer, near=rays.dist_min, far=rays.dist_max).to(dev)
    cam = (torch.arange(N, device=dev) * 2 // N).int()
    jit = jitter.to(dev)

    def run(sel):
        for p in list(nef.parameters()) + [pipe.camera_extrinsics]:
            p.grad = None
        o, d = rays.origins[sel], rays.dirs[sel]
        r = pipe.transform_rays_indexed(o, d, cam[sel]) if pose else pagnerf_amd.Rays(o, d, rays.dist_min, rays.dist_max)
        rb = tracer(nef, channels={"rgb", "depth"}, rays=r, jitter=jit[sel], stage="train")
        ((rb.rgb * G[sel]).sum() + (rb.depth * Gd[sel]).sum()).backward()
        out = {n: p.grad.clone() for n, p in nef.named_parameters() if p.grad is not None}
        if pose:
            out["camera_extrinsics"] = pipe.camera_extrinsics.grad.clone()
        return rb, out
    all_rays = torch.arange(N, device=dev)
    rb_full, g_full = run(all_rays)
    rb_sub, g_sub = run(all_rays[keep.to(dev)])
    assert torch.equal(rb_full.rgb[keep.to(dev)], rb_sub.rgb)
    assert set(g_full) == set(g_sub) and "grid.tables" in g_full
    for name, want in g_sub.items():
        got = g_full[name]
        assert float(want.abs().sum()) > 0, name
        assert T._rel_l2(got.float(), want.float()) < 2e-5, (name, T._rel_l2(got.float(), want.float()))
    # rows of the table that only unused rays touch stay exactly zero
    assert float((g_full["grid.tables"][g_sub["grid.tables"] == 0]).abs().max()) == 0.0


@pytest.mark.parametrize("mode", ["ray", "voxel"])
@pytest.mark.parametrize("layout", ["xcd8", None])
def test_position_gradient_reduced_per_ray_in_the_gather_pass(gpu_device, mode, layout):
    """pag_permuto_encode_bwd_rays (ops._EncodeRays: d origins / d dirs formed per ray inside the position-gradient pass - a segmented scan over each wave's
    lanes, one 6-float slot per (group, wave, ray)) against the per-sample form (pag_permuto_encode_bwd_xyz + pag_ray_sample_grad) on packed samples from the
    real march: ragged rays (occupancy mask), rays without samples, waves that straddle several rays (voxel march: ~10 samples per ray), M % 64 != 0.
    Same table gradient bit for bit; pose gradients to fp32 summation order."""
    import pagnerf_amd
    import test_gpu_parity as T
    from pagnerf_amd import ops
    dev = gpu_device
    N, S = 300, 40
    nef, tracer, rays, occ, jitter = T._make_scene(dev, "bf16", N=N, S=S, cap_log2=12)
    o = rays.origins.clone()
    o[7] = 5.0                                              # a ray that misses the volume: no samples
    g = nef.grid
    res = {}
    for fused in (True, False):
        ops.RAYS_FUSED = fused
        try:
            oo, dd = o.clone().requires_grad_(True), rays.dirs.clone().requires_grad_(True)
            r = pagnerf_amd.Rays(oo, dd, rays.dist_min, rays.dist_max)
            if mode == "ray":
                out = g.raymarch(r, level=None, num_samples=S, raymarch_type="ray", jitter=jitter.to(dev))
            else:
                out = g.raymarch(r, level=None, num_samples=2, raymarch_type="voxel", max_travel=0.9)
            samples = out[2]
            assert hasattr(samples, "_pag_rays") and samples.requires_grad
            M = samples.reshape(-1, 3).shape[0]
            g.tables.grad = None
            feats = g.interpolate_scaled(samples, None, out_dtype=torch.bfloat16 if layout else torch.float32, layout=layout)
            assert (type(feats.grad_fn).__name__ == "_EncodeRaysBackward") == fused
            gen = torch.Generator().manual_seed(3)
            w = torch.randn(feats.shape, generator=gen).to(dev)
            (feats.float() * w).sum().backward()
            res[fused] = (oo.grad.clone(), dd.grad.clone(), g.tables.grad.clone(), M)
        finally:
            ops.RAYS_FUSED = True
    (o1, d1, t1, M1), (o0, d0, t0, M0) = res[True], res[False]
    assert M1 == M0 and M1 % 64 != 0 and float(o0[7].abs().sum()) == 0.0 and float(o1[7].abs().sum()) == 0.0
    assert torch.equal(t1, t0)
    assert float(o0.abs().sum()) > 0 and float(d0.abs().sum()) > 0
    np.testing.assert_allclose(o1.cpu().numpy(), o0.cpu().numpy(), rtol=2e-4, atol=2e-5 * float(o0.abs().max()))
    np.testing.assert_allclose(d1.cpu().numpy(), d0.cpu().numpy(), rtol=2e-4, atol=2e-5 * float(d0.abs().max()))


@pytest.mark.parametrize("mode", ["ray", "voxel"])
def test_view_embedding_gradient_from_per_tile_ray_sums(gpu_device, mode):
    """pag_mlp_bwd_args.dz0_slots: the colour decoder's backward sums dz_0 per (32-sample tile, ray) on the matrix cores and pag_mlp_dz0_slots_sum adds a
    ray's rows - against the [M,64] dz_0 tensor + per-ray segmented sum it replaces: the gradient of the camera extrinsics through the VIEW EMBEDDING alone
    (the position path switched off by detaching the samples is not possible from outside, so the whole pose gradient is compared, to fp32 summation order)
    and every decoder weight gradient bit for bit; ragged rays, tiles that span many rays (voxel march), a ray without samples, M % 32 != 0."""
    import pagnerf_amd
    import test_gpu_parity as T
    from pagnerf_amd import ops
    from pagnerf_amd.ba_pipeline import BAPipeline
    dev = gpu_device
    N, S = 200, 48
    nef, tracer, rays, occ, jitter = T._make_scene(dev, "bf16", N=N, S=S, cap_log2=12)
    if mode == "voxel":
        tracer.raymarch_type, tracer.num_steps, tracer.ray_max_travel = "voxel", 2, 0.9
    o = rays.origins.clone()
    o[11] = 5.0
    views = torch.eye(4).repeat(2, 1, 1)
    views[:, :3, 3] = torch.tensor([[0.01, -0.02, 0.0], [0.0, 0.015, -0.01]])
    pipe = BAPipeline(nef, views, tracer=tracer, near=rays.dist_min, far=rays.dist_max).to(dev)
    cam = (torch.arange(N, device=dev) * 2 // N).int()
    gen = torch.Generator().manual_seed(8)
    G = torch.randn(N, 3, generator=gen).to(dev)
    res = {}
    for slots in (True, False):
        ops.DZ0_SLOTS = slots
        try:
            for p in list(nef.parameters()) + [pipe.camera_extrinsics]:
                p.grad = None
            r = pipe.transform_rays_indexed(o, rays.dirs, cam)
            rb = tracer(nef, channels={"rgb"}, rays=r, jitter=jitter.to(dev), stage="train")
            (rb.rgb * G).sum().backward()
            res[slots] = (pipe.camera_extrinsics.grad.clone(), {n: p.grad.clone() for n, p in nef.named_parameters() if p.grad is not None})
        finally:
            ops.DZ0_SLOTS = True
    (c1, w1), (c0, w0) = res[True], res[False]
    assert float(c0.abs().sum()) > 0
    np.testing.assert_allclose(c1.cpu().numpy(), c0.cpu().numpy(), rtol=5e-4, atol=5e-5 * float(c0.abs().max()))
    for name in w0:
        assert torch.equal(w1[name], w0[name]), name


def test_largest_batch_of_the_dedicated_decoder_kernels(gpu_device):
    """PAG_MLP_FUSED_WIDE_MAX_M (include/pagnerf_hip.h: 2^24 samples - the dedicated decoder kernels address [M,64] bf16 tensors through 32-bit buffer
    descriptors, 2^31 bytes at this size; beyond it the generic kernels run: tests/test_gpu_sizes.py::test_more_than_2pow24_samples_step).
    A trace of exactly 2^24 samples (32768 rays x 512 steps, every sample inside an all-occupied volume), all channels, forward and backward: the rays at
    both ends of the batch and in the middle render as in a trace of those rays alone - bit for bit; the 200-way head to 1e-7 - (what a truncated descriptor would break is exactly the
    tail), gradients are finite and non-zero."""
    import pagnerf_amd
    import test_gpu_parity as T
    dev = gpu_device
    N, S = 32768, 512
    nef, tracer, rays, occ, _ = T._make_scene(dev, "bf16", N=N, S=S, cap_log2=14)
    for grid in (nef.grid, nef.delta_grid):
        grid.blas_init(torch.ones(occ.numel(), dtype=torch.bool))
    gen = torch.Generator().manual_seed(12)
    jit = torch.rand(N, S, generator=gen).to(dev)
    chans = {"rgb", "depth", "semantics", "inst_embedding"}

    def trace(sel, grad):
        r = pagnerf_amd.Rays(rays.origins[sel], rays.dirs[sel], 0.0, 0.5)        # origins within +-0.3, travel <= 0.5: no sample leaves the volume
        with torch.enable_grad() if grad else torch.no_grad():
            return tracer(nef, channels=chans, rays=r, jitter=jit[sel], stage="train")
    full = torch.arange(N, device=dev)
    rb = trace(full, True)
    (rb.rgb.sum() + rb.depth.sum() + rb.semantics.float().square().sum() + rb.inst_embedding.float().square().sum()).backward()
    for name, p in nef.named_parameters():
        if p.grad is not None:
            assert torch.isfinite(p.grad).all(), name
    assert float(nef.grid.tables.grad.abs().sum()) > 0 and float(nef.delta_grid.tables.grad.abs().sum()) > 0
    sub = torch.cat([full[:48], full[N // 2 - 24:N // 2 + 24], full[-48:]])
    rs = trace(sub, False)
    for ch in ("rgb", "depth", "semantics", "inst_embedding"):
        a, b = getattr(rb, ch)[sub], getattr(rs, ch)
        assert float(b.float().abs().sum()) > 0, ch
        if ch == "inst_embedding":      # the wide head's per-ray sums are split over workgroups by batch position: fp32 summation order
            assert float((a.detach().float() - b.float()).abs().max()) < 1e-7
        else:
            assert torch.equal(a, b), (ch, float((a.detach().float() - b.float()).abs().max()))


@pytest.mark.parametrize("mode", ["ray", "voxel"])
def test_wide_head_forward_in_one_pass_matches_the_two_launch_form(gpu_device, mode):
    """pag_mlp_fwd_args.composite (ABI 12: the 200-way head's decoder and its per-ray weighted sum in one launch, every logit and exponential formed once)
    against the statistics launch + pag_head_composite_fwd it replaces: ragged rays (occupancy mask), a ray without samples, rays shorter than a tile
    (voxel march), M % 32 != 0.  Same logits bit for bit; the softmax denominator is summed directly instead of online over the blocks, so the instance
    channel and what the backward rebuilds from (max, 1 / sum) may differ in the last bit: 2e-6 relative on the rendered probabilities, every other
    channel identical, every parameter gradient within 3e-4 (relative L2: bf16 hidden gradients round differently here and there)."""
    import pagnerf_amd
    import test_gpu_parity as T
    from pagnerf_amd import ops
    dev = gpu_device
    N, S = 300, 72
    nef, tracer, rays, occ, jitter = T._make_scene(dev, "bf16", N=N, S=S, cap_log2=12)
    if mode == "voxel":
        tracer.raymarch_type, tracer.num_steps, tracer.ray_max_travel = "voxel", 2, 0.9
    o = rays.origins.clone()
    o[5] = 5.0                                               # misses the volume
    r = pagnerf_amd.Rays(o, rays.dirs, rays.dist_min, rays.dist_max)
    gen = torch.Generator().manual_seed(21)
    G = torch.randn(N, 200, generator=gen).to(dev)
    Gs = torch.randn(N, 6, generator=gen).to(dev)
    seen = {}
    real = ops._call
    res = {}
    min_per_ray = ops.HEAD_FWD_ONCE_MIN_PER_RAY
    for once in (True, False):
        ops.HEAD_FWD_ONCE, ops.HEAD_FWD_ONCE_MIN_PER_RAY = once, 0       # 0: short rays take the one-launch form too (the default keeps it for long rays)
        names = []

        def spy(name, *args):
            names.append(name)
            return real(name, *args)
        ops._call = spy
        try:
            for p in nef.parameters():
                p.grad = None
            rb = tracer(nef, channels={"rgb", "depth", "semantics", "inst_embedding"}, rays=r, jitter=jitter.to(dev), stage="train")
            ((rb.inst_embedding.float() * G).sum() + (rb.semantics.float() * Gs).sum() + rb.rgb.sum()).backward()
            res[once] = (rb, {n: p.grad.clone() for n, p in nef.named_parameters() if p.grad is not None})
            seen[once] = names
        finally:
            ops._call = real
            ops.HEAD_FWD_ONCE, ops.HEAD_FWD_ONCE_MIN_PER_RAY = True, min_per_ray
    assert "pag_head_composite_fwd" not in seen[True] and "pag_head_composite_fwd" in seen[False]
    (rb1, g1), (rb0, g0) = res[True], res[False]
    for ch in ("rgb", "depth", "semantics", "alpha"):
        assert torch.equal(getattr(rb1, ch), getattr(rb0, ch)), ch
    a, b = rb1.inst_embedding.float(), rb0.inst_embedding.float()
    assert float(b.abs().sum()) > 0 and float(a[5].abs().sum()) == 0.0
    assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()) + 1e-9, float((a - b).abs().max())
    assert set(g1) == set(g0)
    for name in g0:
        assert T._rel_l2(g1[name].float(), g0[name].float()) < 3e-4, (name, T._rel_l2(g1[name].float(), g0[name].float()))


@pytest.mark.parametrize("mode", ["ray", "voxel"])
def test_density_and_colour_decoders_in_one_launch_are_bit_identical(gpu_device, mode):
    """pag_mlp_fwd_args.x1_producer (ABI 12: the density decoder evaluated in the colour decoder's launch, its 16 output channels handed on in registers
    in the k order of the standalone launch) against the two launches: rgb, depth, alpha - and with them every gradient - bit for bit, under autograd and
    under no_grad; ragged rays, a ray without samples, M % 32 != 0.  The fused call really is taken (one pag_mlp_fwd less per trace)."""
    import pagnerf_amd
    import test_gpu_parity as T
    from pagnerf_amd import ops
    dev = gpu_device
    N, S = 200, 40
    nef, tracer, rays, occ, jitter = T._make_scene(dev, "bf16", N=N, S=S, cap_log2=12)
    if mode == "voxel":
        tracer.raymarch_type, tracer.num_steps, tracer.ray_max_travel = "voxel", 2, 0.9
    o = rays.origins.clone()
    o[3] = 5.0
    r = pagnerf_amd.Rays(o, rays.dirs, rays.dist_min, rays.dist_max)
    gen = torch.Generator().manual_seed(5)
    G = torch.randn(N, 3, generator=gen).to(dev)
    real = ops._call
    res, calls = {}, {}
    for fused in (True, False):
        ops.CD_FUSED = fused
        names = []

        def spy(name, *args):
            names.append(name)
            return real(name, *args)
        ops._call = spy
        try:
            for p in nef.parameters():
                p.grad = None
            rb = tracer(nef, channels={"rgb", "depth"}, rays=r, jitter=jitter.to(dev), stage="train")
            ((rb.rgb * G).sum() + rb.depth.sum()).backward()
            with torch.no_grad():
                rv = tracer(nef, channels={"rgb", "depth"}, rays=r, jitter=jitter.to(dev), stage="val")
            res[fused] = (rb, rv, {n: p.grad.clone() for n, p in nef.named_parameters() if p.grad is not None})
            calls[fused] = names.count("pag_mlp_fwd")
        finally:
            ops._call = real
            ops.CD_FUSED = True
    assert calls[True] == calls[False] - 2, calls                 # one launch less in the training trace, one less in the no_grad trace
    (rb1, rv1, g1), (rb0, rv0, g0) = res[True], res[False]
    for a, b in ((rb1, rb0), (rv1, rv0), (rb1, rv1)):
        for ch in ("rgb", "depth", "alpha"):
            assert torch.equal(getattr(a, ch).detach(), getattr(b, ch).detach()), ch
    assert float(rb1.rgb.detach().sum()) != 0 and set(g1) == set(g0) and "decoder_density.layers.0.weight" in g1
    for name in g0:
        assert torch.equal(g1[name], g0[name]), name
