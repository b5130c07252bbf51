"""PanopticPackedRFTracer(use_graphs=True): the post-march part of a training trace replayed as HIP graphs over padded static buffers
(pagnerf_amd/graphs.py) against the eager path on the same rays and jitter - forward values bit for bit, every gradient up to the
fp32 summation order of the weight-gradient slabs; a batch that overflows the capacity runs truncated inside the capacity-sized tensors
and the eager path answers; the gradients / outputs handed to the caller obey ordinary ownership rules (accumulation, zero_grad(set_to_none=
False), buffers kept across steps); the split backward (two graphs, N > 1 ranks) equals the single one; no_grad traces never use graphs."""
import collections

import numpy as np
import pytest
import torch

import test_gpu_parity as T
from test_gpu_train_step import ragged_scene, train_loss, hip_leaves

pytestmark = pytest.mark.gpu

CH = {"rgb", "depth", "semantics", "inst_embedding"}


def _targets(N, dev):
    gen = torch.Generator().manual_seed(9)
    return torch.rand(N, 3, generator=gen).to(dev), torch.randint(0, 6, (N,), generator=gen).to(dev), torch.randint(0, 200, (N,), generator=gen).to(dev)


def _step(nef, tracer, rays, jitter, targets, **kw):
    for p in nef.parameters():
        p.grad = None
    rb = tracer(nef, channels=CH, rays=rays, jitter=jitter, stage="train", **kw)
    loss = train_loss(rb.rgb, rb.semantics.float(), rb.inst_embedding.float(), *targets)
    loss.backward()
    torch.cuda.synchronize()
    return rb, loss.detach().clone(), {k: (v.grad.clone() if v.grad is not None else None) for k, v in hip_leaves(nef).items()}


@pytest.mark.parametrize("use", [True, "static"])
@pytest.mark.parametrize("mode", ["ray", "voxel"])
def test_graph_replay_equals_eager(gpu_device, mode, use):
    """use=True: HIP graphs; use="static": the same static padded buffers and optimistic count check with eager launches (what N > 1 runs)."""
    import pagnerf_amd
    dev = gpu_device
    N, S = 96, 32
    nef, tracer, rays, occ, jitter = ragged_scene(dev, "bf16", N=N, S=S)
    kw = {}
    if mode == "voxel":
        tracer.raymarch_type, tracer.num_steps, tracer.ray_max_travel = "voxel", 2, 0.8
        rays.dist_max = 3.0
    jit = jitter.to(dev)
    targets = _targets(N, dev)
    rb_e, loss_e, g_e = _step(nef, tracer, rays, jit, targets)
    gt = pagnerf_amd.PanopticPackedRFTracer(raymarch_type=tracer.raymarch_type, num_steps=tracer.num_steps, bg_color="white",
                                            ray_max_travel=tracer.ray_max_travel, use_graphs=use)
    for it in range(4):                    # 0: eager (learns the count), 1: capture, 2-3: replays
        rb_g, loss_g, g_g = _step(nef, gt, rays, jit, targets)
        for ch in ("rgb", "alpha", "depth", "semantics", "inst_embedding"):
            assert torch.equal(getattr(rb_g, ch), getattr(rb_e, ch)), (it, ch)
        assert torch.equal(rb_g.hit, rb_e.hit) and torch.equal(loss_g, loss_e), it
        for name, want in g_e.items():
            got = g_g[name]
            assert (got is None) == (want is None), name
            if want is not None:
                assert T._rel_l2(got.float(), want.float()) < 1e-5, (it, name, T._rel_l2(got.float(), want.float()))
    assert gt._graphs.captures == (1 if use is True else 0) and gt._graphs.replays == 3 and gt._graphs.overflows == 0
    # different rays through the same graph (same capacity: the count changes, the shapes do not)
    perm = torch.randperm(N, device=dev)
    rays_p = pagnerf_amd.Rays(rays.origins[perm], rays.dirs[perm], rays.dist_min, rays.dist_max)
    tp = tuple(t[perm] for t in targets)
    rb_e2, loss_e2, g_e2 = _step(nef, tracer, rays_p, jit[perm], tp)
    rb_g2, loss_g2, g_g2 = _step(nef, gt, rays_p, jit[perm], tp)
    assert torch.equal(rb_g2.rgb, rb_e2.rgb) and torch.equal(rb_g2.inst_embedding, rb_e2.inst_embedding) and torch.equal(loss_g2, loss_e2)
    assert T._rel_l2(g_g2["delta_grid.tables"].float(), g_e2["delta_grid.tables"].float()) < 1e-5
    # no_grad / validation traces never take the graph path
    before = (gt._graphs.captures, gt._graphs.replays)
    with torch.no_grad():
        rb_v = gt(nef, channels=CH, rays=rays, jitter=jit, stage="val")
    assert (gt._graphs.captures, gt._graphs.replays) == before and torch.equal(rb_v.rgb, rb_e.rgb.detach())


@pytest.mark.parametrize("mode", ["ray", "voxel"])
def test_graph_replay_equals_eager_with_the_one_launch_wide_head(gpu_device, mode):
    """The same comparison with the 200-way head's forward in its one-launch form (pag_mlp_fwd_args.composite; by default only taken for long rays) on
    these short rays: the padded batch's filler samples past the last pack belong to no ray - the launch gives them statistics that rebuild to
    probability 0 and a zero hidden row, so the replayed backward (which reads them scaled by a zero weight) matches the eager one."""
    from pagnerf_amd import ops
    keep = ops.HEAD_FWD_ONCE_MIN_PER_RAY
    ops.HEAD_FWD_ONCE_MIN_PER_RAY = 0
    try:
        test_graph_replay_equals_eager(gpu_device, mode, True)
    finally:
        ops.HEAD_FWD_ONCE_MIN_PER_RAY = keep


def test_graph_training_tracks_eager_training(gpu_device):
    """Twenty Adam steps on fresh random rays with and without graphs from the same initial state: the losses follow each other
    (differences come only from the summation order of the weight gradients)."""
    import pagnerf_amd
    dev = gpu_device
    N, S = 256, 48
    losses = {}
    for use in (False, True, "static"):
        nef, tracer, rays, occ, jitter = T._make_scene(dev, "bf16", N=N, S=S, cap_log2=12)
        tr = pagnerf_amd.PanopticPackedRFTracer(raymarch_type="ray", num_steps=S, bg_color="white", use_graphs=use)
        opt = torch.optim.Adam(nef.parameters(), lr=1e-3, eps=1e-15)
        gen = torch.Generator().manual_seed(1)
        out = []
        for it in range(20):
            o = ((torch.rand(N, 3, generator=gen) - 0.5) * 0.6).to(dev)
            d = torch.nn.functional.normalize(torch.randn(N, 3, generator=gen), dim=-1).to(dev)
            jit = torch.rand(N, S, generator=gen).to(dev)
            gtc = torch.rand(N, 3, generator=gen).to(dev)
            opt.zero_grad(set_to_none=True)
            rb = tr(nef, channels={"rgb", "semantics"}, rays=pagnerf_amd.Rays(o, d, 0.0, 2.0), jitter=jit, stage="train")
            loss = 10.0 * torch.abs(rb.rgb - gtc).mean() - 0.1 * torch.log(rb.semantics[:, 0] + 1e-27).mean()
            loss.backward()
            opt.step()
            out.append(float(loss.detach()))
        losses[use] = out
        if use:
            assert tr._graphs.replays >= 15, (tr._graphs.replays, tr._graphs.captures, tr._graphs.overflows)
    np.testing.assert_allclose(losses[True], losses[False], rtol=2e-3)
    np.testing.assert_allclose(losses["static"], losses[False], rtol=2e-3)


def test_graph_states_are_bounded(gpu_device):
    """Every configuration (channel set, precision, tables, ...) owns march buffers and two graphs; a runner keeps at most MAX_STATES of
    them and drops the least recently used - a long training that changes configuration now and then must not accumulate captures."""
    import pagnerf_amd
    from pagnerf_amd.graphs import GraphRunner
    dev = gpu_device
    nef, _, rays, occ, jitter = T._make_scene(dev, "bf16", N=64, S=16, cap_log2=10)
    nef.train()
    tr = pagnerf_amd.PanopticPackedRFTracer(raymarch_type="ray", num_steps=16, bg_color="white", use_graphs=True)
    sets = [{"rgb"}, {"rgb", "depth"}, {"rgb", "semantics"}, {"rgb", "inst_embedding"}, {"rgb", "depth", "semantics"},
            {"rgb", "depth", "semantics", "inst_embedding"}]
    for ch in sets:
        for _ in range(3):          # eager (observes the count), capture, replay
            rb = tr(nef, channels=ch, rays=rays, stage="train")
            rb.rgb.sum().backward()
    g = tr._graphs
    assert g.captures == len(sets) and len(g.states) == GraphRunner.MAX_STATES
    # the oldest configuration was evicted: it is observed and captured again, results as the eager path gives them
    before = g.captures
    for _ in range(3):
        rb = tr(nef, channels=sets[0], rays=rays, stage="train")
    assert g.captures == before + 1 and len(g.states) == GraphRunner.MAX_STATES
    eager = pagnerf_amd.PanopticPackedRFTracer(raymarch_type="ray", num_steps=16, bg_color="white")
    torch.manual_seed(1)
    j = torch.rand(64, 16, device=dev)
    a = tr(nef, channels=sets[0], rays=rays, stage="train", jitter=j).rgb
    b = eager(nef, channels=sets[0], rays=rays, stage="train", jitter=j).rgb
    assert torch.equal(a, b)


def _half_outside(rays, dev):
    """The same rays with every second one moved out of the volume (no samples): a batch about half the size under the SAME graph key."""
    import pagnerf_amd
    o, d = rays.origins.clone(), rays.dirs.clone()
    o[::2] = torch.tensor([3.0, 3.0, 3.0], device=dev)
    d[::2] = torch.nn.functional.normalize(torch.tensor([1.0, 1.0, 1.0], device=dev), dim=0)
    return pagnerf_amd.Rays(o, d, rays.dist_min, rays.dist_max)


@pytest.mark.parametrize("use", [True, "static"])
@pytest.mark.parametrize("mode", ["ray", "voxel"])
def test_overflowing_batch_stays_inside_the_capacity(gpu_device, mode, use, monkeypatch):
    """A capacity captured on small batches, then a batch twice as large through the same configuration: the launches queued before
    the host learns the count walk the CLAMPED pack table (no pack reaches past the capacity), `overflows` counts it, the eager path
    answers with the eager values, and the next steps (capacity grown) replay again."""
    import pagnerf_amd
    from pagnerf_amd import graphs
    monkeypatch.setattr(graphs, "GRANULE", 64)
    dev = gpu_device
    N, S = 96, 32
    nef, tracer, rays, occ, jitter = ragged_scene(dev, "bf16", N=N, S=S)
    if mode == "voxel":
        tracer.raymarch_type, tracer.num_steps, tracer.ray_max_travel = "voxel", 2, 0.8
        rays.dist_max = 3.0
    jit = jitter.to(dev)
    targets = _targets(N, dev)
    small = _half_outside(rays, dev)
    gt = pagnerf_amd.PanopticPackedRFTracer(raymarch_type=tracer.raymarch_type, num_steps=tracer.num_steps, bg_color="white",
                                            ray_max_travel=tracer.ray_max_travel, use_graphs=use)
    for _ in range(3):                      # eager, capture (or first static step), replay - all on the small batch
        _step(nef, gt, small, jit, targets)
    g = gt._graphs
    st = next(iter(g.states.values()))
    cap = st.cap
    rb_e, loss_e, g_e = _step(nef, tracer, rays, jit, targets)
    M = int(st.buf.pack_start[N])           # still the small batch's table
    assert M <= cap and g.overflows == 0
    rb_o, loss_o, g_o = _step(nef, gt, rays, jit, targets)
    M_big = int(st.buf.pack_start[N])
    assert M_big > cap, (M_big, cap)        # the batch really overflowed the capacity the launches ran with
    assert g.overflows == 1
    ps, pc = st.buf.pack_start, st.buf.pack_start_c
    assert int(pc.max()) == cap and torch.equal(pc, ps.clamp(max=cap))
    for ch in ("rgb", "alpha", "depth", "semantics", "inst_embedding"):
        assert torch.equal(getattr(rb_o, ch), getattr(rb_e, ch)), ch
    assert torch.equal(loss_o, loss_e)
    for name, want in g_e.items():
        if want is not None:
            assert T._rel_l2(g_o[name].float(), want.float()) < 1e-5, name
    # the capacity has grown with the observed count: the same batch now fits and is replayed
    before = g.replays
    rb_2, loss_2, _ = _step(nef, gt, rays, jit, targets)
    assert st.cap >= M_big and g.overflows == 1 and g.replays == before + 1
    assert torch.equal(rb_2.rgb, rb_e.rgb) and torch.equal(loss_2, loss_e)
    rb_3, loss_3, _ = _step(nef, gt, rays, jit, targets)
    assert torch.equal(rb_3.inst_embedding, rb_e.inst_embedding) and torch.equal(loss_3, loss_e)
    assert len(st.buckets) <= graphs.MAX_BUCKETS


def _fresh_tracers(tracer, **kw):
    import pagnerf_amd
    mk = lambda use: pagnerf_amd.PanopticPackedRFTracer(raymarch_type=tracer.raymarch_type, num_steps=tracer.num_steps, bg_color="white",
                                                        ray_max_travel=tracer.ray_max_travel, use_graphs=use, **kw)
    return mk(False), mk(True)


def _warm(nef, gt, rays, jit, targets):
    for _ in range(3):                      # eager (learns the count), capture, replay
        _step(nef, gt, rays, jit, targets)


def test_graph_gradients_accumulate_like_eager(gpu_device):
    """Two traces (different rays) of one captured configuration, each followed by backward(), then ONE optimiser step: p.grad = g1 + g2
    as the eager path gives it - the first backward's p.grad (which autograd adopted from the capture's static buffer) must survive
    the second replay."""
    dev = gpu_device
    N, S = 96, 32
    nef, tracer, rays, occ, jitter = ragged_scene(dev, "bf16", N=N, S=S)
    jit = jitter.to(dev)
    targets = _targets(N, dev)
    et, gt = _fresh_tracers(tracer)
    _warm(nef, gt, rays, jit, targets)
    perm = torch.randperm(N, device=dev)
    import pagnerf_amd
    rays2 = pagnerf_amd.Rays(rays.origins[perm], rays.dirs[perm], rays.dist_min, rays.dist_max)
    t2 = tuple(t.flip(0) for t in targets)

    def two_traces(tr):
        for p in nef.parameters():
            p.grad = None
        for r, j, t in ((rays, jit, targets), (rays2, jit[perm], t2)):
            rb = tr(nef, channels=CH, rays=r, jitter=j, stage="train")
            train_loss(rb.rgb, rb.semantics.float(), rb.inst_embedding.float(), *t).backward()
        torch.cuda.synchronize()
        return {k: v.grad.clone() for k, v in hip_leaves(nef).items() if v.grad is not None}
    want = two_traces(et)
    got = two_traces(gt)
    assert gt._graphs.overflows == 0 and gt._graphs.captures == 1
    assert set(got) == set(want)
    for name in want:
        assert T._rel_l2(got[name].float(), want[name].float()) < 1e-5, (name, T._rel_l2(got[name].float(), want[name].float()))
    # and it is NOT 2 x g2 (what an aliased p.grad would hold): g1 != g2 here
    _, _, g2 = _step(nef, et, rays2, jit[perm], t2)
    assert T._rel_l2(got["delta_grid.tables"].float(), 2 * g2["delta_grid.tables"].float()) > 1e-2


def test_graph_zero_grad_in_place_tracks_eager(gpu_device):
    """Five optimiser steps with zero_grad(set_to_none=False) - p.grad keeps the tensor autograd adopted from the capture - and a
    GradScaler-style in-place unscale of p.grad between backward and step: the losses follow the eager path to 1e-5, every step's
    gradient to 1e-4 and the accumulated parameter UPDATE to 1e-2 (an update is ~1e-4 .. 1e-5 of a parameter, i.e. it carries the
    parameter's own fp32 rounding at the 1e-3 level; an aliased p.grad gave 2 x the gradient: error ~1).  Plain SGD on purpose: Adam's update is invariant to the scale of the gradient (a doubled gradient - what an aliased p.grad
    produced - would pass) and turns fp32 summation-order noise on near-zero entries into +-lr steps."""
    dev = gpu_device
    N, S = 96, 32
    finals = {}
    for use in (False, True):
        nef, tracer, rays, occ, jitter = ragged_scene(dev, "fp32", N=N, S=S)       # seeded: the same initial state both times
        jit = jitter.to(dev)
        targets = _targets(N, dev)
        tr = _fresh_tracers(tracer)[1 if use else 0]
        opt = torch.optim.SGD(nef.parameters(), lr=1e-4)
        if use:
            _warm(nef, tr, rays, jit, targets)          # no optimiser step in there: the parameters are still the initial ones
        init = {k: v.detach().clone() for k, v in hip_leaves(nef).items()}
        losses, grads = [], []
        for it in range(5):
            opt.zero_grad(set_to_none=False)
            rb = tr(nef, channels=CH, rays=rays, jitter=jit, stage="train")
            loss = train_loss(rb.rgb, rb.semantics.float(), rb.inst_embedding.float(), *targets) * 4.0
            loss.backward()
            for p in nef.parameters():      # what GradScaler.unscale_ does: in place on p.grad
                if p.grad is not None:
                    p.grad.mul_(0.25)
            grads.append({k: v.grad.clone() for k, v in hip_leaves(nef).items()})
            opt.step()
            losses.append(float(loss.detach()))
        finals[use] = (losses, {k: v.detach() - init[k] for k, v in hip_leaves(nef).items()}, grads)
        if use:
            assert tr._graphs.replays >= 5 and tr._graphs.overflows == 0
    np.testing.assert_allclose(finals[True][0], finals[False][0], rtol=2e-5)
    assert finals[False][0][-1] != finals[False][0][0]
    for name, want in finals[False][1].items():
        assert float(want.abs().max()) > 0, name
        assert T._rel_l2(finals[True][1][name].float(), want.float()) < 1e-2, (name, T._rel_l2(finals[True][1][name].float(), want.float()))
        for it in range(5):
            e = T._rel_l2(finals[True][2][it][name].float(), finals[False][2][it][name].float())
            assert e < 1e-4, (it, name, e)


def test_graph_outputs_are_owned_by_the_caller(gpu_device):
    """The RenderBuffer of step k is bit-unchanged after step k + 1 (different rays), and so is a gradient tensor the caller kept."""
    import pagnerf_amd
    dev = gpu_device
    N, S = 96, 32
    nef, tracer, rays, occ, jitter = ragged_scene(dev, "bf16", N=N, S=S)
    jit = jitter.to(dev)
    targets = _targets(N, dev)
    _, gt = _fresh_tracers(tracer)
    _warm(nef, gt, rays, jit, targets)
    rb1, _, g1 = _step(nef, gt, rays, jit, targets)
    kept = {ch: getattr(rb1, ch).clone() for ch in ("rgb", "alpha", "depth", "semantics", "inst_embedding", "hit")}
    perm = torch.randperm(N, device=dev)
    rays2 = pagnerf_amd.Rays(rays.origins[perm], rays.dirs[perm], rays.dist_min, rays.dist_max)
    rb2, _, _ = _step(nef, gt, rays2, jit[perm], tuple(t[perm] for t in targets))
    assert not torch.equal(rb2.rgb, kept["rgb"])
    for ch, want in kept.items():
        assert torch.equal(getattr(rb1, ch), want), ch
    # a second backward over a trace whose forward graph has been replayed since must fail loudly, not return stale gradients
    rb_a = gt(nef, channels=CH, rays=rays, jitter=jit, stage="train")
    rb_b = gt(nef, channels=CH, rays=rays2, jitter=jit[perm], stage="train")
    with pytest.raises(RuntimeError, match="replayed again"):
        rb_a.rgb.sum().backward()
    rb_b.rgb.sum().backward()


def test_split_backward_graphs_equal_the_single_graph(gpu_device):
    """graph_split=True (what N > 1 ranks run): the panoptic heads' backward - which completes the delta table's gradient - is one graph,
    the rest another, each behind its own autograd node; a post-accumulate hook on the delta table fires BEFORE the main table's
    gradient exists.  Values are bit-equal to the single backward graph."""
    dev = gpu_device
    N, S = 96, 32
    nef, tracer, rays, occ, jitter = ragged_scene(dev, "bf16", N=N, S=S)
    jit = jitter.to(dev)
    targets = _targets(N, dev)
    _, single = _fresh_tracers(tracer)
    _, split = _fresh_tracers(tracer, graph_split=True)
    _warm(nef, single, rays, jit, targets)
    _warm(nef, split, rays, jit, targets)
    order = []
    leaves = hip_leaves(nef)
    hooks = [leaves[n].register_post_accumulate_grad_hook(lambda p, n=n: order.append(n)) for n in ("delta_grid.tables", "grid.tables")]
    try:
        rb_s, loss_s, g_s = _step(nef, single, rays, jit, targets)
        order.clear()
        rb_p, loss_p, g_p = _step(nef, split, rays, jit, targets)
    finally:
        for h in hooks:
            h.remove()
    gr = next(iter(split._graphs.states.values()))
    graphed = next(iter(gr.buckets.values()))
    assert len(graphed.groups) == 2
    assert order == ["delta_grid.tables", "grid.tables"], order
    assert torch.equal(loss_p, loss_s)
    for ch in ("rgb", "alpha", "depth", "semantics", "inst_embedding"):
        assert torch.equal(getattr(rb_p, ch), getattr(rb_s, ch)), ch
    for name, want in g_s.items():
        got = g_p[name]
        assert (got is None) == (want is None), name
        if want is not None:
            assert torch.equal(got, want), name


def test_graph_key_follows_requires_grad(gpu_device):
    """A parameter frozen / unfrozen after a capture is a different configuration: the unfrozen parameter receives its gradient."""
    dev = gpu_device
    N, S = 96, 32
    nef, tracer, rays, occ, jitter = ragged_scene(dev, "bf16", N=N, S=S)
    jit = jitter.to(dev)
    targets = _targets(N, dev)
    _, gt = _fresh_tracers(tracer)
    nef.delta_grid.tables.requires_grad_(False)
    _warm(nef, gt, rays, jit, targets)
    assert nef.delta_grid.tables.grad is None and gt._graphs.captures == 1
    nef.delta_grid.tables.requires_grad_(True)
    _warm(nef, gt, rays, jit, targets)
    assert gt._graphs.captures == 2
    _, _, g = _step(nef, gt, rays, jit, targets)
    et, _ = _fresh_tracers(tracer)
    _, _, ge = _step(nef, et, rays, jit, targets)
    assert g["delta_grid.tables"] is not None and T._rel_l2(g["delta_grid.tables"].float(), ge["delta_grid.tables"].float()) < 1e-5


@pytest.mark.parametrize("use", [True, "static", "default"])
@pytest.mark.parametrize("mode", ["ray", "voxel"])
def test_pose_optimisation_through_the_graph_path(gpu_device, mode, use, monkeypatch):
    """configs/bup20/best.yaml runs EVERY step with learnable extrinsics (optimize_extrinsics, extrinsics_epoch_end 900 > epochs 800:
    pc_nerf/trainer.py:308, pc_nerf/ba_pipeline.py:85-92): rays that require a gradient take the graph path too.  Against the eager
    path on the same rays and jitter: forward bit for bit, d loss / d camera_extrinsics (through d origins / d dirs: the main grid's
    position gradient, the per-ray sums of pag_ray_sample_grad, the view embedding) and every parameter gradient to 1e-5."""
    import pagnerf_amd
    from pagnerf_amd.ba_pipeline import BAPipeline
    dev = gpu_device
    N, S, C = 96, 32, 3
    nef, tracer, rays, occ, jitter = ragged_scene(dev, "bf16", N=N, S=S)
    if mode == "voxel":
        tracer.raymarch_type, tracer.num_steps, tracer.ray_max_travel = "voxel", 2, 0.8
    jit = jitter.to(dev)
    targets = _targets(N, dev)
    views = torch.eye(4).repeat(C, 1, 1)
    views[:, :3, 3] = torch.tensor([[0.01, -0.02, 0.0], [0.0, 0.015, -0.01], [-0.02, 0.0, 0.02]])
    far = 3.0 if mode == "voxel" else rays.dist_max

    def run(tr, steps):
        pipe = BAPipeline(nef, views, tracer=tr, near=rays.dist_min, far=far).to(dev)
        cam = (torch.arange(N, device=dev) * C // N)
        out = []
        for it in range(steps):
            for p in list(nef.parameters()) + [pipe.camera_extrinsics]:
                p.grad = None
            world = pipe.transform_rays_indexed(rays.origins, rays.dirs, cam)        # camera-frame rays = the scene's rays (identity rotations)
            assert world.origins.requires_grad and world.dirs.requires_grad
            rb = tr(nef, channels=CH, rays=world, jitter=jit, stage="train")
            loss = train_loss(rb.rgb, rb.semantics.float(), rb.inst_embedding.float(), *targets) + rb.depth.sum() * 0.01
            loss.backward()
            torch.cuda.synchronize()
            grads = {k: (v.grad.clone() if v.grad is not None else None) for k, v in hip_leaves(nef).items()}
            grads["camera_extrinsics"] = pipe.camera_extrinsics.grad.clone()
            out.append((rb, loss.detach().clone(), grads))
        return out
    (rb_e, loss_e, g_e), = run(tracer, 1)
    assert float(g_e["camera_extrinsics"].abs().sum()) > 0
    if use == "default":      # the product default: no use_graphs argument, no PAG_GRAPHS in the environment -> HIP graphs
        monkeypatch.delenv("PAG_GRAPHS", raising=False)
        gt = pagnerf_amd.PanopticPackedRFTracer(raymarch_type=tracer.raymarch_type, num_steps=tracer.num_steps, bg_color="white", ray_max_travel=tracer.ray_max_travel)
        assert gt.use_graphs is True
        use = True
    else:
        gt = pagnerf_amd.PanopticPackedRFTracer(raymarch_type=tracer.raymarch_type, num_steps=tracer.num_steps, bg_color="white",
                                                ray_max_travel=tracer.ray_max_travel, use_graphs=use)
    for it, (rb_g, loss_g, g_g) in enumerate(run(gt, 4)):                  # 0: eager (learns the count), 1: capture, 2-3: replays
        for ch in ("rgb", "alpha", "depth", "semantics", "inst_embedding"):
            assert torch.equal(getattr(rb_g, ch), getattr(rb_e, ch)), (it, ch)
        assert torch.equal(loss_g, loss_e), it
        for name, want in g_e.items():
            got = g_g[name]
            assert (got is None) == (want is None), name
            if want is not None:
                assert T._rel_l2(got.float(), want.float()) < 1e-5, (it, name, T._rel_l2(got.float(), want.float()))
    assert gt._graphs.captures == (1 if use is True else 0) and gt._graphs.replays == 3 and gt._graphs.overflows == 0
    # the same tracer on rays WITHOUT a gradient afterwards: a different configuration (its own capture), same values
    world = BAPipeline(nef, views, tracer=None, near=rays.dist_min, far=far).to(dev).transform_rays_indexed(rays.origins, rays.dirs, torch.arange(N, device=dev) * C // N)
    plain = pagnerf_amd.Rays(world.origins.detach(), world.dirs.detach(), rays.dist_min, far)
    for _ in range(3):
        rb_p = gt(nef, channels=CH, rays=plain, jitter=jit, stage="train")
        rb_p.rgb.sum().backward()
    assert torch.equal(rb_p.rgb, rb_e.rgb) and gt._graphs.captures == (2 if use is True else 0)


def test_default_constructed_tracer_replays_graphs_in_a_reference_style_loop(gpu_device, monkeypatch):
    """The product default: a tracer built without `use_graphs` (as the reference's YAML builds it) in the reference's loop - zero_grad(set_to_none=True),
    pipeline(..., stage='train'), GradScaler backward / step (pc_nerf/trainer.py:426-435,582-584) - replays HIP graphs from its third step on; a
    validation trace in between (no_grad, stage='val') takes the eager path and does not disturb the capture; the training curve follows the eager tracer's."""
    import pagnerf_amd
    monkeypatch.delenv("PAG_GRAPHS", raising=False)
    dev = gpu_device
    N, S = 256, 48
    curves = {}
    for use in (None, False):
        nef, _, rays, occ, jitter = T._make_scene(dev, "bf16", N=N, S=S, cap_log2=12)
        tr = pagnerf_amd.PanopticPackedRFTracer(raymarch_type="ray", num_steps=S, bg_color="white") if use is None else \
            pagnerf_amd.PanopticPackedRFTracer(raymarch_type="ray", num_steps=S, bg_color="white", use_graphs=False)
        assert tr.use_graphs is (True if use is None else False)
        pipe = pagnerf_amd.Pipeline(nef, tr)
        opt = pagnerf_amd.optim.Adam(nef.parameters(), lr=1e-3, eps=1e-15)
        scaler = torch.amp.GradScaler("cuda", init_scale=128.0)
        gen = torch.Generator().manual_seed(1)
        out = []
        for it in range(10):
            jit = torch.rand(N, S, generator=gen).to(dev)
            gtc = torch.rand(N, 3, generator=gen).to(dev)
            opt.zero_grad(set_to_none=True)
            rb = pipe(rays=rays, lod_idx=None, channels=["rgb", "semantics", "inst_embedding", "depth"], stage="train", jitter=jit)
            loss = 10.0 * torch.abs(rb.rgb - gtc).mean() - 0.1 * torch.log(rb.semantics[:, 0] + 1e-27).mean() - torch.log(rb.inst_embedding[:, 3] + 1e-27).mean()
            scaler.scale(loss).backward()
            scaler.step(opt)
            scaler.update()
            out.append(float(loss.detach()))
            if it == 5:
                with torch.no_grad():
                    val = pipe(rays=rays, lod_idx=None, channels=["rgb"])            # trace()'s default stage is 'val'
                assert val.rgb.shape == (N, 3)
        curves[use] = out
        if use is None:
            g = tr._graphs
            assert g is not None and g.captures == 1 and g.replays >= 8 and g.overflows == 0, (g.captures, g.replays, g.overflows)
    np.testing.assert_allclose(curves[None], curves[False], rtol=2e-3)
