"""pagnerf_amd.PanopticNeF - the reference's BASE field (pc_nerf/panoptic_nef.py:20,253-363: one grid, sem_detach / inst_detach,
inst_direct_pos) - against golden g9 (channels and reference-autograd gradients of the reference class on a HashGridTorch grid) and,
through the tracer's fused panoptic path on a permutohedral grid, against torch autograd over the oracle chain."""
import numpy as np
import pytest
import torch

import test_gpu_parity as T
from conftest import golden, table_from_seed

pytestmark = pytest.mark.gpu

CFG = dict(dd=(True, True, False), ll=(False, False, False), ld=(False, True, False), pos=(True, True, True))


def _g9_nef(g, tag, precision, dev):
    import pagnerf_amd
    sd, idt, direct = CFG[tag]
    Lv, log2T = int(g["L"]), int(g["log2T"])
    nef = pagnerf_amd.PanopticNeF(grid_type="HashGridTorch", feature_dim=2, num_lods=Lv, num_classes=6, num_instances=200, sem_num_layers=2,
                                  sem_softmax=True, inst_num_layers=1, inst_softmax=True, sem_detach=sd, inst_detach=idt, inst_direct_pos=direct,
                                  panoptic_features_type="position" if direct else None, codebook_bitwidth=log2T, precision=precision)
    assert not hasattr(nef, "delta_grid") and nef.get_nef_type() == "panoptic_nef"
    nef.grid.init_from_resolutions([int(g["res"][0])] * (Lv - 1) + [int(g["res"][-1])])
    nef.grid.tables.data.copy_(torch.from_numpy(table_from_seed(int(g["seed_main"]), (Lv, 2 ** log2T, 2), "normal") * np.float32(0.5)))
    wt = tag if tag in ("dd", "pos") else "dd"
    for short in ("density", "color", "semantics", "inst"):
        dec = getattr(nef, "decoder_" + short)
        for i, lin in enumerate(list(dec.layers) + [dec.lout]):
            lin.weight.data.copy_(torch.from_numpy(g[f"{wt}_decoder_{short}_w{i}"]))
            lin.bias.data.copy_(torch.from_numpy(g[f"{wt}_decoder_{short}_b{i}"]))
    return nef.to(dev)


@pytest.mark.parametrize("tag", ["dd", "ll", "ld", "pos"])
def test_g9_base_nef_against_reference_golden(gpu_device, tag):
    """fp32 path: channels to 1e-5 and every gradient (grid table, decoders) of the golden's linear functional to 1e-4 of the
    reference's autograd; bf16 path: channels to 2e-2.  With a detach flag off the panoptic term reaches grid.tables."""
    dev = gpu_device
    g = golden("g9_base_nef.npz")
    direct = CFG[tag][2]
    chans = {"density", "rgb", "inst_embedding"} | (set() if direct else {"semantics"})
    coords, ray_d = torch.from_numpy(g["coords"]).to(dev), torch.from_numpy(g["ray_d"]).to(dev)
    for precision in ("fp32", "bf16"):
        nef = _g9_nef(g, tag, precision, dev)
        out = nef(coords=coords, ray_d=ray_d, pidx=None, lod_idx=None, channels=chans)
        tol = dict(rtol=1e-5, atol=2e-6) if precision == "fp32" else dict(rtol=3e-2, atol=3e-2)
        for c in chans:
            assert tuple(out[c].shape) == tuple(g[f"{tag}_{c}"].shape), (c, out[c].shape)       # 'pos': decoder_inst(coords) is [batch, num_samples, I]
            np.testing.assert_allclose(out[c].detach().float().cpu().numpy(), g[f"{tag}_{c}"], err_msg=f"{precision} {c}", **tol)
        if precision != "fp32":
            continue
        loss = sum((out[c].float() * torch.from_numpy(g["G_" + c]).to(dev).reshape(out[c].shape)).sum() for c in chans)
        loss.backward()
        want = torch.from_numpy(g[f"{tag}_dtables"])
        got = nef.grid.tables.grad.float().cpu()
        assert float((got - want).abs().max()) < 1e-4 * float(want.abs().max()), (tag, float((got - want).abs().max()), float(want.abs().max()))
        for short in ("density", "color", "semantics", "inst"):
            dec = getattr(nef, "decoder_" + short)
            for i, lin in enumerate(list(dec.layers) + [dec.lout]):
                key = f"{tag}_decoder_{short}_dw{i}"
                if key not in g.files:
                    assert lin.weight.grad is None or float(lin.weight.grad.abs().max()) == 0.0, key
                    continue
                wg = torch.from_numpy(g[key])
                assert float((lin.weight.grad.float().cpu() - wg).abs().max()) < 1e-4 * float(wg.abs().max()) + 1e-7, key
                bg = torch.from_numpy(g[f"{tag}_decoder_{short}_db{i}"])
                assert float((lin.bias.grad.float().cpu() - bg).abs().max()) < 1e-4 * float(wg.abs().max()) + 1e-7, key
        if tag in ("ll", "ld"):
            base = torch.from_numpy(g["dd_dtables"])
            assert float((got - base).abs().max()) > 1e-2 * float(base.abs().max())            # the panoptic term did reach the grid


def _oracle_base_step(nef, rays, occ, jitter, S, targets, sd, idt):
    """loss + leaf gradients of the all-channel train step of the BASE field from torch autograd over the oracle chain."""
    from oracle import permuto_encode as op, decoders as od, render as orr
    from test_gpu_train_step import train_loss
    o, d = rays.origins.cpu(), rays.dirs.cpu()
    N = o.shape[0]
    ridx, pidx, samples, depths, deltas, boundary = orr.raymarch_ray(o, d, rays.dist_min, rays.dist_max, S, jitter, occ, nef.grid.blas_level)
    xyz = samples[:, 0].numpy()
    xyz = op.half_round(xyz) if nef.grid.half_coords else xyz
    grid = nef.grid
    tab = grid.tables.detach().float().cpu().clone().requires_grad_(True)
    _, idx, bary = op.permuto_encode(xyz, tab.detach().numpy(), grid.random_shift_per_level.cpu().numpy(), grid.scale_factors(grid.resolutions).numpy())
    idx_t, bary_t = torch.from_numpy(idx.astype(np.int64)), torch.from_numpy(bary)
    feats = torch.cat([(tab[l][idx_t[l]] * bary_t[l][..., None]).sum(1) for l in range(tab.shape[0])], -1)
    leaves, params = {"grid.tables": tab}, {}
    for short in ("density", "color", "semantics", "inst"):
        W, b = getattr(nef, "decoder_" + short).weights()
        Wc = [w.detach().float().cpu().clone().requires_grad_(True) for w in W]
        bc = [v.detach().float().cpu().clone().requires_grad_(True) for v in b]
        params[short] = (Wc, bc)
        for i in range(len(Wc)):
            leaves["decoder_%s.W%d" % (short, i)], leaves["decoder_%s.b%d" % (short, i)] = Wc[i], bc[i]
    out = od.nef_forward_base(feats, d[ridx], params, {"rgb", "semantics", "inst_embedding"}, lod_weights=nef.lod_weights, sem_detach=sd, inst_detach=idt)
    comp = orr.composite(N, ridx, boundary, out["density"], deltas, depths=depths, rgb=out["rgb"], bg_color="white")
    pan = orr.composite(N, ridx, boundary, out["density"].detach(), deltas, semantics=out["semantics"], inst=out["inst_embedding"], bg_color="white")
    loss = train_loss(comp["rgb"], pan["semantics"], pan["inst_embedding"], *targets)
    loss.backward()
    return loss.detach(), {k: v.grad for k, v in leaves.items()}


@pytest.mark.parametrize("flags", [(True, True), (False, False), (False, True)])
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_base_nef_train_step_through_the_tracer(gpu_device, precision, flags):
    """The all-channel train step of PanopticNeF on a permutohedral grid through PanopticPackedRFTracer (bf16: each head fused with its
    compositing, the heads paired when they read the same tensor) against torch autograd over the oracle chain: with a detach flag off
    the semantic / instance NLL terms add to the main table's gradient."""
    import pagnerf_amd
    from test_gpu_train_step import train_loss
    dev = gpu_device
    sd, idt = flags
    N, S, L_perm, cap_log2 = 96, 32, 24, 10
    torch.manual_seed(0)
    nef = pagnerf_amd.PanopticNeF(grid_type="PermutoGrid", feature_dim=2, num_lods=L_perm, num_classes=6, num_instances=200, sem_num_layers=2,
                                  sem_softmax=True, inst_num_layers=1, inst_softmax=True, sem_detach=sd, inst_detach=idt,
                                  capacity_log_2=cap_log2, coarsest_scale=1.0, finest_scale=1e-4, blas_level=5, precision=precision)
    gen = torch.Generator().manual_seed(0)
    nef.grid.init_from_scales(random_shift=torch.randn(L_perm, 3, generator=gen) * 10, tables=torch.randn(L_perm, 2 ** cap_log2, 2, generator=gen) * 0.3)
    nef = nef.to(dev)
    tracer = pagnerf_amd.PanopticPackedRFTracer(raymarch_type="ray", num_steps=S, bg_color="white")
    o = (torch.rand(N, 3, generator=gen) - 0.5) * 0.6
    d = torch.nn.functional.normalize(torch.randn(N, 3, generator=gen), dim=-1)
    o[5] = torch.tensor([3.0, 3.0, 3.0])
    d[5] = torch.nn.functional.normalize(torch.tensor([1.0, 1.0, 1.0]), dim=0)
    rays = pagnerf_amd.Rays(o.to(dev), d.to(dev), dist_min=0.0, dist_max=2.0)
    occ = torch.rand(32, 32, 32, generator=gen) > 0.35
    nef.grid.blas_init(occ.reshape(-1))
    jitter = torch.rand(N, S, generator=gen)
    targets = (torch.rand(N, 3, generator=gen), torch.randint(0, 6, (N,), generator=gen), torch.randint(0, 200, (N,), generator=gen))
    ref_loss, ref = _oracle_base_step(nef, rays, occ, jitter, S, targets, sd, idt)
    rb = tracer(nef, channels={"rgb", "depth", "semantics", "inst_embedding"}, rays=rays, jitter=jitter.to(dev), stage="train")
    loss = train_loss(rb.rgb, rb.semantics.float(), rb.inst_embedding.float(), *[t.to(dev) for t in targets])
    loss.backward()
    rel = abs(float(loss.detach()) - float(ref_loss)) / abs(float(ref_loss))
    assert rel < (1e-4 if precision == "fp32" else 3e-2), (float(loss.detach()), float(ref_loss))
    leaves = {"grid.tables": nef.grid.tables}
    for short in ("density", "color", "semantics", "inst"):
        W, b = getattr(nef, "decoder_" + short).weights()
        for i in range(len(W)):
            leaves["decoder_%s.W%d" % (short, i)], leaves["decoder_%s.b%d" % (short, i)] = W[i], b[i]
    worst = {}
    for name, p in leaves.items():
        got, want = p.grad.float().cpu(), ref[name]
        if precision == "fp32":
            worst[name] = float((got - want).abs().max()) / (float(want.abs().max()) + 1e-20)
        else:
            worst[name] = T._rel_l2(got, want)
    lim = 2e-3 if precision == "fp32" else 6e-2
    assert max(worst.values()) < lim, worst
    # the panoptic terms alone: they reach the grid exactly when a head reads live features
    for p in leaves.values():
        p.grad = None
    rb = tracer(nef, channels={"rgb", "semantics", "inst_embedding"}, rays=rays, jitter=jitter.to(dev), stage="train")
    F = torch.nn.functional
    pan_only = 0.1 * F.nll_loss(torch.log(rb.semantics.float() + 1e-27), targets[1].to(dev)) \
        + 1000.0 * F.nll_loss(torch.log(rb.inst_embedding.float() + 1e-27), targets[2].to(dev))
    pan_only.backward()
    touched = nef.grid.tables.grad is not None and float(nef.grid.tables.grad.abs().max()) > 0
    assert touched == (not (sd and idt)), (flags, touched)
    dens_touched = any(p.grad is not None and float(p.grad.abs().max()) > 0 for n, p in leaves.items() if n.startswith(("decoder_density", "decoder_color")))
    assert not dens_touched          # the compositing weights of the panoptic channels carry no gradient (tracer :148-155)


@pytest.mark.parametrize("split", [False, True])
def test_base_nef_through_the_graph_path(gpu_device, split):
    """PanopticNeF with both detach flags off behind use_graphs=True: the panoptic outputs and the colour / density outputs BOTH carry a
    gradient for grid.tables - with the split backward each graph returns its share and autograd sums them.  Forward bit for bit and
    gradients to the summation order of the weight-gradient slabs against the eager path, also with p.grad kept across steps.
    One legitimate difference of the split form: eager / single-graph autograd adds the two bf16 input gradients of the grid (from the
    density decoder and from the heads) BEFORE the one encode backward - a bf16 addition, 2^-9 relative - while the split form runs the
    encode backward on each and adds the fp32 table gradients: the main table agrees to 4e-3 there, everything else to 1e-5."""
    import pagnerf_amd
    from test_gpu_train_step import train_loss
    dev = gpu_device
    N, S, L_perm, cap_log2 = 96, 32, 24, 10
    torch.manual_seed(0)
    nef = pagnerf_amd.PanopticNeF(grid_type="PermutoGrid", feature_dim=2, num_lods=L_perm, num_classes=6, num_instances=200, sem_num_layers=2,
                                  sem_softmax=True, inst_num_layers=1, inst_softmax=True, sem_detach=False, inst_detach=False,
                                  capacity_log_2=cap_log2, coarsest_scale=1.0, finest_scale=1e-4, blas_level=5, precision="bf16")
    gen = torch.Generator().manual_seed(0)
    nef.grid.init_from_scales(random_shift=torch.randn(L_perm, 3, generator=gen) * 10, tables=torch.randn(L_perm, 2 ** cap_log2, 2, generator=gen) * 0.3)
    nef = nef.to(dev)
    o = ((torch.rand(N, 3, generator=gen) - 0.5) * 0.6).to(dev)
    d = torch.nn.functional.normalize(torch.randn(N, 3, generator=gen), dim=-1).to(dev)
    rays = pagnerf_amd.Rays(o, d, dist_min=0.0, dist_max=2.0)
    nef.grid.blas_init((torch.rand(32, 32, 32, generator=gen) > 0.35).reshape(-1))
    jit = torch.rand(N, S, generator=gen).to(dev)
    targets = tuple(t.to(dev) for t in (torch.rand(N, 3, generator=gen), torch.randint(0, 6, (N,), generator=gen), torch.randint(0, 200, (N,), generator=gen)))
    CH = {"rgb", "depth", "semantics", "inst_embedding"}

    def step(tr, keep_grads=False):
        if not keep_grads:
            for p in nef.parameters():
                p.grad = None
        rb = tr(nef, channels=CH, rays=rays, jitter=jit, stage="train")
        loss = train_loss(rb.rgb, rb.semantics.float(), rb.inst_embedding.float(), *targets)
        loss.backward()
        torch.cuda.synchronize()
        return rb, {n: p.grad.clone() for n, p in nef.named_parameters() if p.grad is not None}
    eager = pagnerf_amd.PanopticPackedRFTracer(raymarch_type="ray", num_steps=S, bg_color="white")
    graph = pagnerf_amd.PanopticPackedRFTracer(raymarch_type="ray", num_steps=S, bg_color="white", use_graphs=True, graph_split=split)
    rb_e, g_e = step(eager)
    for it in range(4):
        rb_g, g_g = step(graph)
        for ch in ("rgb", "alpha", "depth", "semantics", "inst_embedding"):
            assert torch.equal(getattr(rb_g, ch), getattr(rb_e, ch)), (it, ch)
        assert set(g_g) == set(g_e)
        for n in g_e:
            tol = 4e-3 if (split and n == "grid.tables") else 1e-5
            assert T._rel_l2(g_g[n].float(), g_e[n].float()) < tol, (it, n, T._rel_l2(g_g[n].float(), g_e[n].float()))
    gr = next(iter(next(iter(graph._graphs.states.values())).buckets.values()))
    assert len(gr.groups) == (2 if split else 1)
    if split:       # both graphs hold a gradient for the main table
        assert sum(any(p is nef.grid.tables for p in g.params) for g in gr.groups) == 2
    # accumulation over two traces (p.grad kept): 2 x the single-step gradient
    _, g2 = step(graph, keep_grads=True)
    for n in g_e:
        assert T._rel_l2(g2[n].float(), 2 * g_e[n].float()) < (4e-3 if (split and n == "grid.tables") else 1e-5), n
