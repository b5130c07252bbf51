"""Forward-only rendering (the reference's other caller of the path: pc_nerf/trainer.py:637-649 batch_render, :943-999 validate):
what a no_grad trace launches, and that the chunked image render equals the per-chunk traces."""
import ctypes

import pytest
import torch

import test_gpu_parity as T

pytestmark = pytest.mark.gpu

CH = {"rgb", "depth", "semantics", "inst_embedding"}


def _spy_fwd_args(fn):
    """Run fn() with a spy on ops._call: -> list of (entry point name, snapshot of the pag_mlp_fwd argument struct or None)."""
    from pagnerf_amd import ops
    seen = []
    real = ops._call

    def spy(name, *args):
        snap = None
        if name == "pag_mlp_fwd":
            a = args[0]._obj
            chain = [a] + ([a.pair.contents] if a.pair else []) + ([a.x1_producer.contents] if a.x1_producer else [])      # decoders riding in this launch
            snap = [dict(out_dim=s.out_dim, out=s.out, hidden=[s.hidden_save[i] for i in range(2)], stats=s.softmax_stats,
                         col0=s.x1_col0_relu, n_layers=s.n_layers, composite=bool(s.composite)) for s in chain]
        seen.append((name, snap))
        return real(name, *args)
    ops._call = spy
    try:
        out = fn()
    finally:
        ops._call = real
    return out, seen


@pytest.mark.parametrize("mode", ["ray", "voxel"])
def test_inference_trace_writes_no_training_only_tensors(gpu_device, mode):
    """Under torch.no_grad() the decoder launches keep nothing for a backward: no hidden activations, no softmax statistics - except
    what the statistics-only wide head hands to its own compositing launch (last hidden layer + statistics, consumed in the same trace:
    the [M,200] probability tensor they replace is never written) - and no backward / weight-gradient entry point runs.  The same
    trace with gradients enabled gives the same buffers bit for bit."""
    dev = gpu_device
    nef, tracer, rays, occ, jitter = T._make_scene(dev, "bf16", N=128, S=48)
    if mode == "voxel":
        tracer.raymarch_type, tracer.num_steps, tracer.ray_max_travel = "voxel", 2, 0.8
    jit = jitter.to(dev)[:, :tracer.num_steps] if mode == "ray" else None
    kw = dict(jitter=jit) if jit is not None else {}

    def run():
        return tracer(nef, channels=CH, rays=rays, stage="val", **kw)
    with torch.no_grad():
        rb, seen = _spy_fwd_args(run)
    names = [n for n, _ in seen]
    assert not any(("bwd" in n or "wgrad" in n) for n in names), names
    fwd = [s for n, snaps in seen if n == "pag_mlp_fwd" for s in snaps]
    assert len(fwd) == 4, [(s["out_dim"], s["n_layers"]) for s in fwd]            # density, colour, instance (statistics only), semantic
    for s in fwd:
        if s["out_dim"] == 200:                 # the wide head: no [M,200] output; last hidden layer + statistics feed pag_head_composite_fwd
            assert s["out"] is None and s["stats"] is not None and s["hidden"][s["n_layers"] - 2] is not None
            assert all(h is None for i, h in enumerate(s["hidden"]) if i != s["n_layers"] - 2), s
        else:
            assert s["hidden"] == [None, None] and s["stats"] is None, s
    # ... in the decoder's own launch (pag_mlp_fwd_args.composite, ABI 12) or, without it, in pag_head_composite_fwd
    wide = [s for s in fwd if s["out_dim"] == 200]
    assert len(wide) == 1 and (wide[0]["composite"] != ("pag_head_composite_fwd" in names))
    rb_g, seen_g = _spy_fwd_args(run)           # gradients enabled: same values
    for ch in ("rgb", "alpha", "depth", "semantics", "inst_embedding", "hit"):
        assert torch.equal(getattr(rb, ch), getattr(rb_g, ch).detach()), ch
    assert all(not t.requires_grad for t in (rb.rgb, rb.semantics, rb.inst_embedding))


@pytest.mark.parametrize("mode", ["ray", "voxel"])
def test_batch_render_image_equals_per_chunk_traces(gpu_device, mode):
    """pagnerf_amd.batch_render (trainer.py:637-649): a 60 x 40 'image' in chunks of 700 rays equals the concatenation of the
    per-chunk traces and - rays are independent - one trace of all rays, bit for bit (dense occupancy: no jitter-free march needed,
    the jitter is fixed per chunk through the generator).  batch_render marches pack i + 1 on a second stream while pack i is shaded
    (PanopticPackedRFTracer.render_packs): the plain loop beside it is the reference's; 'voxel' = the march every validation after the first prune runs."""
    import pagnerf_amd
    dev = gpu_device
    nef, tracer, rays, occ, _ = T._make_scene(dev, "bf16", N=2400, S=32)
    if mode == "voxel":
        tracer.raymarch_type, tracer.num_steps, tracer.ray_max_travel = "voxel", 2, 0.8
        rays.dist_max = 3.0
    pipe = pagnerf_amd.Pipeline(nef, tracer)
    chunk = 700
    with torch.no_grad():
        torch.manual_seed(3)
        rb = pagnerf_amd.batch_render(pipe, rays, channels=sorted(CH), render_batch=chunk)
        torch.manual_seed(3)
        parts = [pipe(rays=r, lod_idx=None, channels=sorted(CH)) for r in rays.split(chunk)]
    for ch in ("rgb", "alpha", "depth", "semantics", "inst_embedding", "hit"):
        want = torch.cat([getattr(p, ch) for p in parts], 0)
        got = getattr(rb, ch)
        assert got.shape[0] == 2400 and torch.equal(got, want), ch
