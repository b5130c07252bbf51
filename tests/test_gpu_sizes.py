"""GPU checks at the batch sizes of BASELINE.json configs[3] (6 images x 4096 rays x 512 samples, ba_pipeline pose optimisation) and
configs[4] (the 131 072-ray x 64-sample per-GPU shard of the 1 M-ray batch, fp32 and fp16 tables), plus one batch beyond 2^24 packed
samples (32-bit offsets, the documented switch from the dedicated to the generic decoder kernels - DESIGN 4.3b).

The oracle cannot run these sizes, so they are checked through size-independent properties:
  * determinism and ray-permutation equivariance, bitwise (no atomics, fixed summation orders);
  * a contiguous ray block rendered alone equals its rows of the full render, and that block (65 536 samples) matches the CPU oracle;
  * table gradients: binned (atomic-free) == per-vertex atomics == additive over ray blocks;
  * pose optimisation: an image's camera gradient from the 24 576-ray step equals the gradient of that image rendered alone.
"""
import numpy as np
import pytest
import torch

import test_gpu_parity as T

pytestmark = pytest.mark.gpu


def _downward_rays(n, dev, seed, far=1.9):
    """bench.make_rays(): downward-looking rays that stay inside [-1,1]^3 (every sample survives a dense occupancy grid)."""
    import pagnerf_amd
    g = torch.Generator().manual_seed(seed)
    o = torch.cat([(torch.rand(n, 2, generator=g) - 0.5) * 0.6, torch.full((n, 1), 0.95)], 1)
    d = torch.nn.functional.normalize(torch.cat([(torch.rand(n, 2, generator=g) - 0.5) * 0.7, -torch.ones(n, 1)], 1), dim=-1)
    return pagnerf_amd.Rays(o.to(dev), d.to(dev), dist_min=0.0, dist_max=far)


def _scene(dev, N, S, table_dtype=torch.float32, grid="permuto", seed=0):
    import pagnerf_amd
    torch.manual_seed(seed)
    common = dict(feature_dim=2, num_classes=6, num_instances=200, sem_num_layers=1, sem_softmax=True, inst_num_layers=2, inst_softmax=True,
                  panoptic_features_type="delta", blas_level=7, precision="bf16", table_dtype=table_dtype)
    gen = torch.Generator().manual_seed(seed)
    if grid == "permuto":
        nef = pagnerf_amd.PanopticDeltaNeF(grid_type="PermutoGrid", num_lods=24, capacity_log_2=18, delta_capacity_log_2=18,
                                           coarsest_scale=1.0, finest_scale=1e-4, **common)
        for g in (nef.grid, nef.delta_grid):
            g.init_from_scales(random_shift=torch.randn(24, 3, generator=gen) * 10, tables=torch.randn(24, 2 ** 18, 2, generator=gen) * 0.05)
    else:
        nef = pagnerf_amd.PanopticDeltaNeF(grid_type="HashGridTorch", num_lods=16, codebook_bitwidth=19, **common)
        for g in (nef.grid, nef.delta_grid):
            g.init_from_resolutions([16] * 15 + [1024])
            g.tables.data.copy_((torch.randn(16, 2 ** 19, 2, generator=gen) * 0.05).to(table_dtype))
    nef = nef.to(dev)
    tracer = pagnerf_amd.PanopticPackedRFTracer(raymarch_type="ray", num_steps=S, bg_color="white")
    return nef, tracer, _downward_rays(N, dev, seed + 1)


@pytest.mark.parametrize("grid,table_dtype", [("permuto", torch.float32), ("permuto", torch.float16), ("hash", torch.float32)])
def test_config4_shard_render_properties_and_oracle_block(gpu_device, grid, table_dtype):
    import pagnerf_amd
    dev = gpu_device
    N, S = 131072, 64
    nef, tracer, rays = _scene(dev, N, S, table_dtype, grid)
    assert nef.grid.tables.dtype == table_dtype
    jit = torch.rand(N, S, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    chans = {"rgb", "depth"}
    with torch.no_grad():
        rb = tracer(nef, channels=chans, rays=rays, jitter=jit)
        rb2 = tracer(nef, channels=chans, rays=rays, jitter=jit)
        assert torch.equal(rb.rgb, rb2.rgb) and torch.equal(rb.depth, rb2.depth)                           # determinism
        assert torch.isfinite(rb.rgb).all() and float(rb.alpha.min()) >= 0 and float(rb.alpha.max()) <= 1 + 1e-5
        perm = torch.randperm(N, device=dev, generator=torch.Generator(device=dev).manual_seed(4))
        rp = pagnerf_amd.Rays(rays.origins[perm], rays.dirs[perm], rays.dist_min, rays.dist_max)
        rbp = tracer(nef, channels=chans, rays=rp, jitter=jit[perm])
        assert torch.equal(rbp.rgb, rb.rgb[perm]) and torch.equal(rbp.depth, rb.depth[perm])             # ray-permutation equivariance
        lo, nb = 70001, 1024                                                                               # a block that starts mid-tile
        blk = pagnerf_amd.Rays(rays.origins[lo:lo + nb], rays.dirs[lo:lo + nb], rays.dist_min, rays.dist_max)
        rbb = tracer(nef, channels={"rgb", "depth", "semantics", "inst_embedding"}, rays=blk, jitter=jit[lo:lo + nb])
        assert torch.equal(rbb.rgb, rb.rgb[lo:lo + nb]) and torch.equal(rbb.depth, rb.depth[lo:lo + nb])   # a block alone == its rows
    if grid != "permuto":
        return
    # the 65 536 samples of that block against the CPU oracle (fp32 chain; bf16 path: 2e-2 absolute on post-activation values)
    occ = torch.ones(128, 128, 128, dtype=torch.bool)
    comp, _, _ = T._oracle_render(nef, blk, occ, jit[lo:lo + nb].cpu(), S, {"rgb", "depth", "semantics", "inst_embedding"})
    for ch in ("rgb", "alpha", "depth", "semantics", "inst_embedding"):
        np.testing.assert_allclose(getattr(rbb, ch).float().cpu().numpy(), comp[ch].numpy(), rtol=0, atol=2e-2, err_msg=ch)


@pytest.mark.parametrize("grid", ["permuto", "hash"])
def test_config4_shard_table_gradients_binned_atomic_additive(gpu_device, grid):
    """Encode backward at M = 131 072 x 64 = 8 388 608 samples, production layout (bf16 XCD8 gradient): the binned kernels against the
    per-vertex atomics, and additivity over two sample blocks."""
    from pagnerf_amd import ops
    dev = gpu_device
    nef, tracer, rays = _scene(dev, 64, 64, grid=grid)
    g = nef.grid
    M = 131072 * 64
    gen = torch.Generator(device=dev).manual_seed(5)
    xyz = torch.rand(M, 3, device=dev, generator=gen) * 1.9 - 0.95
    xyz = (xyz.reshape(-1, 64, 3) * torch.tensor([1.0, 1.0, 0.0], device=dev)
           + torch.linspace(-0.9, 0.9, 64, device=dev)[None, :, None] * torch.tensor([0.0, 0.0, 1.0], device=dev)).reshape(M, 3)   # 64 ordered samples per "ray"
    go = (torch.randn(8, M, 8, device=dev, generator=gen) * 0.1).bfloat16()
    Lv = g.num_lods
    pad = torch.tensor([c < 0 for c in ops.xcd8_columns(Lv, 2)], device=dev).reshape(8, 8)
    go = torch.where(pad[:, None, :], torch.zeros_like(go), go)

    def table_grad(x, gpiece, algo):
        ops.BWD_ALGO = algo
        try:
            t = g.tables.detach().clone().requires_grad_(True)
            out = ops.encode(x, t, g._spec, layout="xcd8", half_coords=g.rounds_coords())
            out.backward(gpiece)
            return t.grad.float()
        finally:
            ops.BWD_ALGO = "binned"
    full = table_grad(xyz, go, "binned")
    again = table_grad(xyz, go, "binned")
    assert torch.equal(full, again)                                                                       # bitwise reproducible
    half = M // 2 + 12345
    parts = table_grad(xyz[:half], go[:, :half].contiguous(), "binned") + table_grad(xyz[half:], go[:, half:].contiguous(), "binned")
    assert T._rel_l2(parts, full) < 1e-4, T._rel_l2(parts, full)
    # the XCD8 layout always takes the binned kernels; the atomic reference runs on the same gradient as a strided [M, C] tensor
    cols = ops.xcd8_columns(Lv, 2)
    flat = go.permute(1, 0, 2).reshape(M, 64)
    strided = torch.zeros(M, Lv * 2, device=dev, dtype=torch.bfloat16)
    for p_, c in enumerate(cols):
        if c >= 0:
            strided[:, c] = flat[:, p_]
    ops.BWD_ALGO = "atomic"
    try:
        t = g.tables.detach().clone().requires_grad_(True)
        ops.encode(xyz, t, g._spec, out_dtype=torch.bfloat16, half_coords=g.rounds_coords()).backward(strided)
        atomic = t.grad.float()
    finally:
        ops.BWD_ALGO = "binned"
    assert T._rel_l2(full, atomic) < 2e-4, T._rel_l2(full, atomic)      # the fp32 atomic sum carries ~eps * sqrt(adds per row) itself


def test_more_than_2pow24_samples_step(gpu_device):
    """131 072 rays x 160 samples = 20 971 520 packed samples (> 2^24: 32-bit sample offsets x 8-byte pieces pass 2^27 B, the decoders
    leave the dedicated kernels - DESIGN 4.3b): one full all-channel train step.  A ray block rendered alone (dedicated kernels) must
    match its rows of the big render; the big step's gradients must equal the sum over two halves of the rays."""
    import pagnerf_amd
    dev = gpu_device
    N, S = 131072, 160
    nef, tracer, rays = _scene(dev, N, S)
    gen = torch.Generator(device=dev).manual_seed(8)
    jit = torch.rand(N, S, device=dev, generator=gen)
    gt = torch.rand(N, 3, device=dev, generator=gen)
    sem_gt = torch.randint(0, 6, (N,), device=dev, generator=gen)
    inst_gt = torch.randint(0, 200, (N,), device=dev, generator=gen)
    chans = {"rgb", "depth", "semantics", "inst_embedding"}
    params = [p for p in nef.parameters()]

    def step(lo, hi):
        for p in params:
            p.grad = None
        r = pagnerf_amd.Rays(rays.origins[lo:hi], rays.dirs[lo:hi], rays.dist_min, rays.dist_max)
        rb = tracer(nef, channels=chans, rays=r, jitter=jit[lo:hi], stage="train")
        F = torch.nn.functional
        loss = torch.abs(rb.rgb - gt[lo:hi]).sum() * 1e-3 \
            + F.nll_loss(torch.log(rb.semantics.float() + 1e-27), sem_gt[lo:hi], reduction="sum") * 1e-4 \
            + F.nll_loss(torch.log(rb.inst_embedding.float() + 1e-27), inst_gt[lo:hi], reduction="sum") * 1e-4
        loss.backward()
        torch.cuda.synchronize()
        return rb, [p.grad.float().clone() if p.grad is not None else None for p in params]
    rb, g_full = step(0, N)
    assert rb.rgb.shape == (N, 3) and all(torch.isfinite(x).all() for x in g_full if x is not None)
    with torch.no_grad():
        lo, nb = 99999, 2048
        blk = pagnerf_amd.Rays(rays.origins[lo:lo + nb], rays.dirs[lo:lo + nb], rays.dist_min, rays.dist_max)
        rbb = tracer(nef, channels=chans, rays=blk, jitter=jit[lo:lo + nb])
        for ch in ("rgb", "depth", "semantics", "inst_embedding"):
            a, b = getattr(rbb, ch).float(), getattr(rb, ch).detach().float()[lo:lo + nb]
            assert float((a - b).abs().max()) < 2e-2, ch
    _, g_a = step(0, N // 2)
    _, g_b = step(N // 2, N)
    names = [n for n, _ in nef.named_parameters()]
    for n, f, a, b in zip(names, g_full, g_a, g_b):
        if f is None:
            continue
        assert T._rel_l2(a + b, f) < 2e-2, (n, T._rel_l2(a + b, f))


def test_config3_pose_opt_step_per_image_additivity(gpu_device):
    """6 images x 4096 rays x 512 samples through BAPipeline (12 582 912 packed samples, all channels): the gradient of image c's camera
    row from the whole step equals the gradient of that image rendered alone, the anchor frame gets none (ba_pipeline.py:56-60), the
    table gradients add up over the images."""
    import pagnerf_amd
    dev = gpu_device
    C, per, S = 6, 4096, 512
    N = C * per
    nef, tracer, _ = _scene(dev, 64, S)
    gen = torch.Generator().manual_seed(7)
    views = torch.eye(4).repeat(C, 1, 1)
    ang = (torch.rand(C, generator=gen) - 0.5) * 0.3
    views[:, 0, 0], views[:, 0, 1], views[:, 1, 0], views[:, 1, 1] = torch.cos(ang), -torch.sin(ang), torch.sin(ang), torch.cos(ang)
    views[:, :3, 3] = torch.cat([(torch.rand(C, 2, generator=gen) - 0.5) * 0.2, torch.full((C, 1), -0.95)], 1)
    pipe = pagnerf_amd.BAPipeline(nef, views, tracer=tracer, anchor_frame_idxs=[0], near=0.0, far=1.9).to(dev)
    o = torch.zeros(N, 3, device=dev)
    d = torch.cat([(torch.rand(N, 2, generator=gen) - 0.5) * 0.7, -torch.ones(N, 1)], 1).to(dev)
    cam = (torch.arange(N) // per).to(dev)
    gdev = torch.Generator(device=dev).manual_seed(9)
    jit = torch.rand(N, S, device=dev, generator=gdev)
    G = torch.randn(N, 3, device=dev, generator=gdev)
    chans = {"rgb", "depth", "semantics", "inst_embedding"}
    leaves = [pipe.camera_extrinsics, nef.grid.tables, nef.delta_grid.tables]

    def step(lo, hi):
        for p in leaves:
            p.grad = None
        rays = pipe.transform_rays_indexed(o[lo:hi], d[lo:hi], cam[lo:hi])
        rb = tracer(nef, channels=chans, rays=rays, jitter=jit[lo:hi], stage="train")
        ((rb.rgb * G[lo:hi]).sum() * 1e-2 + rb.depth.sum() * 1e-3 + rb.semantics[:, 1].sum() * 1e-2 + rb.inst_embedding[:, 5].sum()).backward()
        torch.cuda.synchronize()
        return [p.grad.float().clone() for p in leaves]
    full = step(0, N)
    assert float(full[0][0].abs().sum()) == 0.0 and float(full[0][1:].abs().max(dim=1).values.min()) > 1e-4    # anchor masked, the others live
    acc = [torch.zeros_like(x) for x in full]
    for c in range(C):
        part = step(c * per, (c + 1) * per)
        others = torch.ones(C, dtype=torch.bool)
        others[c] = False
        assert float(part[0][others.to(dev)].abs().sum()) == 0.0                                              # an image moves its own camera only
        if c > 0:
            assert T._rel_l2(part[0][c], full[0][c]) < 1e-2, (c, T._rel_l2(part[0][c], full[0][c]))
        for a, p_ in zip(acc, part):
            a += p_
    for name, a, f in zip(("camera_extrinsics", "grid.tables", "delta_grid.tables"), acc, full):
        assert T._rel_l2(a, f) < 1e-2, (name, T._rel_l2(a, f))
