"""PanopticPackedRFTracer(use_graphs=True): the post-march part of a training trace replayed as HIP graphs over padded static buffers
(pagnerf_amd/graphs.py) against the eager path on the same rays and jitter - forward values bit for bit, every gradient up to the
fp32 summation order of the weight-gradient slabs; capacity overflow falls back to the eager path; no_grad traces never use graphs."""
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
    # a capacity that the batch overflows: the replay is discarded, the eager path answers (same values)
    st = next(iter(gt._graphs.states.values()))
    st.counts = collections.deque([64], maxlen=8)
    st.buckets = {k: v for k, v in st.buckets.items()}
    rb_o, loss_o, _ = _step(nef, gt, rays, jit, targets)
    assert torch.equal(rb_o.rgb, rb_e.rgb) and torch.equal(loss_o, loss_e)
    # no_grad / validation traces never take the graph path
    before = (gt._graphs.captures, gt._graphs.replays)
    with torch.no_grad():
        rb_v = gt(nef, channels=CH, rays=rays, jitter=jit, stage="val")
    assert (gt._graphs.captures, gt._graphs.replays) == before and torch.equal(rb_v.rgb, rb_e.rgb.detach())


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
