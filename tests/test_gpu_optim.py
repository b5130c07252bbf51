"""pagnerf_amd.optim.Adam (pag_adam_step) against torch.optim.Adam - the optimiser the reference builds (config_parser.py:667-673,
eps = 1e-15) over the parameter groups of pc_nerf/trainer.py:268-286 and steps through GradScaler (:583)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _params(dev, seed=0):
    g = torch.Generator().manual_seed(seed)
    shapes = [(24, 4096, 2), (24, 4099, 2), (64, 48), (64,), (200, 64), (3,), (1,), (70001,)]        # table-like (aligned and odd), decoder-like, tiny, odd tail
    return [torch.nn.Parameter((torch.randn(s, generator=g) * 0.3).to(dev)) for s in shapes]


def _grads(params, step, touched_fraction=0.3):
    g = torch.Generator().manual_seed(100 + step)
    out = []
    for p in params:
        gr = torch.randn(p.shape, generator=g) * (10.0 ** float(torch.randint(-6, 3, (1,), generator=g)))
        if p.numel() > 1000:               # tables: most rows untouched (exact zeros) - with eps = 1e-15 they must not move
            gr = gr * (torch.rand(p.shape[:-1] + (1,) if p.dim() > 1 else p.shape, generator=g) < touched_fraction)
        out.append(gr.to(p.device))
    return out


@pytest.mark.parametrize("weight_decay", [0.0, 1e-2])
def test_adam_matches_torch_over_a_trajectory(gpu_device, weight_decay):
    """25 steps, two parameter groups (lr x 100 for the 'grid' group as trainer.py:272-281), eps = 1e-15, gradients spanning nine
    decades with exact zeros on most table rows: every parameter and both moments stay within a few ulp of torch.optim.Adam's
    single-tensor implementation; never-touched rows do not move at all."""
    import pagnerf_amd
    dev = gpu_device
    pa, pb = _params(dev), _params(dev)
    groups = lambda ps: [dict(params=ps[:2] + ps[7:], lr=0.1), dict(params=ps[2:7], lr=1e-3)]
    oa = pagnerf_amd.optim.Adam(groups(pa), eps=1e-15, weight_decay=weight_decay)
    ob = torch.optim.Adam(groups(pb), eps=1e-15, weight_decay=weight_decay, foreach=False, fused=False)
    never = [torch.ones(p.shape, dtype=torch.bool, device=dev) for p in pa]
    init = [p.detach().clone() for p in pa]
    for step in range(25):
        for p, q, g, nv in zip(pa, pb, _grads(pa, step), never):
            p.grad, q.grad = g.clone(), g.clone()
            nv &= g == 0
        oa.step()
        ob.step()
    for i, (p, q) in enumerate(zip(pa, pb)):
        scale = float(q.detach().abs().max())
        assert float((p.detach() - q.detach()).abs().max()) <= 4e-6 * scale, (i, float((p.detach() - q.detach()).abs().max()), scale)
        sa, sb = oa.state[p], ob.state[q]
        assert float(sa["step"]) == float(sb["step"]) == 25.0
        for k in ("exp_avg", "exp_avg_sq"):
            ref = sb[k]
            assert float((sa[k] - ref).abs().max()) <= 4e-6 * float(ref.abs().max()) + 1e-30, (i, k)
        if weight_decay == 0.0 and bool(never[i].any()):
            assert torch.equal(p.detach()[never[i]], init[i][never[i]])          # 0 / (0 + 1e-15) = 0: untouched rows stay bit for bit


def test_adam_interoperates_with_torch_state_and_gradscaler(gpu_device):
    """state_dict() of either class loads into the other and the trajectories continue together; GradScaler.step() drives it as it
    drives torch.optim.Adam (unscale, inf check, skipped step on inf); groups the kernel does not cover (fp16 parameters, amsgrad)
    take torch's implementation."""
    import pagnerf_amd
    dev = gpu_device
    pa, pb = _params(dev, 1), _params(dev, 1)
    oa = pagnerf_amd.optim.Adam(pa, lr=1e-2, eps=1e-15)
    ob = torch.optim.Adam(pb, lr=1e-2, eps=1e-15)
    for step in range(3):
        for p, q, g in zip(pa, pb, _grads(pa, step)):
            p.grad, q.grad = g.clone(), g.clone()
        oa.step()
        ob.step()
    # swap the optimiser states through state_dict and continue
    sa, sb = copy.deepcopy(oa.state_dict()), copy.deepcopy(ob.state_dict())
    oa2 = pagnerf_amd.optim.Adam(pa, lr=1e-2, eps=1e-15)
    ob2 = torch.optim.Adam(pb, lr=1e-2, eps=1e-15)
    oa2.load_state_dict(sb)
    ob2.load_state_dict(sa)
    scaler_a, scaler_b = torch.amp.GradScaler("cuda", init_scale=1024.0), torch.amp.GradScaler("cuda", init_scale=1024.0)
    for sc in (scaler_a, scaler_b):
        sc.scale(torch.ones(1, device=dev))        # what `scaler.scale(loss)` (trainer.py:582) does first: creates the scale tensor
    for step in range(3, 7):
        for p, q, g in zip(pa, pb, _grads(pa, step)):
            p.grad, q.grad = g.clone() * 1024.0, g.clone() * 1024.0
            if step == 5 and p.numel() == 3:
                p.grad[0] = float("inf")
                q.grad[0] = float("inf")
        before = pa[0].detach().clone()
        scaler_a.step(oa2)
        scaler_a.update()
        scaler_b.step(ob2)
        scaler_b.update()
        if step == 5:
            assert torch.equal(pa[0].detach(), before)                          # inf found: the step is skipped, as with torch's Adam
    for p, q in zip(pa, pb):
        assert float((p.detach() - q.detach()).abs().max()) <= 4e-6 * float(q.detach().abs().max())
    assert float(oa2.state[pa[0]]["step"]) == float(ob2.state[pb[0]]["step"]) == 6.0
    # fallback groups
    h = torch.nn.Parameter(torch.randn(1000, device=dev).half())
    f = torch.nn.Parameter(torch.randn(1000, device=dev))
    oc = pagnerf_amd.optim.Adam([dict(params=[h], eps=1e-4), dict(params=[f], amsgrad=True)], lr=1e-2)
    h0, f0 = h.detach().clone(), f.detach().clone()
    h.grad, f.grad = torch.ones_like(h), torch.ones_like(f)
    oc.step()
    assert not torch.equal(h.detach(), h0) and not torch.equal(f.detach(), f0) and "max_exp_avg_sq" in oc.state[f]


def test_decoupled_weight_decay_and_hooks_on_fallback_groups(gpu_device):
    """torch >= 2.7: Adam(decoupled_weight_decay=True) is AdamW (`p *= 1 - lr * wd`); pag_adam_step implements the L2 form, so such a group
    with a non-zero decay must take torch's implementation (the reference puts weight_decay on the grid groups, trainer.py:272-281) - the
    trajectory follows torch.optim.Adam(decoupled_weight_decay=True), not the L2 one.  Step hooks fire ONCE per step() although the
    fallback groups are stepped by torch's own step() from inside ours."""
    import pagnerf_amd
    dev = gpu_device
    pa, pb, pc = _params(dev, 2), _params(dev, 2), _params(dev, 2)
    kw = dict(lr=1e-2, eps=1e-15, weight_decay=1e-1)
    ob = torch.optim.Adam(pb, decoupled_weight_decay=True, foreach=False, fused=False, **kw)       # constructed first: marks torch's step as hooked
    oa = pagnerf_amd.optim.Adam([dict(params=pa[:2], decoupled_weight_decay=True), dict(params=pa[2:], weight_decay=0.0)], **kw)
    oc = torch.optim.Adam(pc, foreach=False, fused=False, **kw)                                     # L2 form: must NOT be what the tables follow
    assert not oa._group_ok(oa.param_groups[0]) and oa._group_ok(oa.param_groups[1])
    fired = []
    oa.register_step_pre_hook(lambda *a: fired.append("pre"))
    oa.register_step_post_hook(lambda *a: fired.append("post"))
    for step in range(5):
        for p, q, r, g in zip(pa, pb, pc, _grads(pa, step)):
            p.grad, q.grad, r.grad = g.clone(), g.clone(), g.clone()
        oa.step()
        ob.step()
        oc.step()
    assert fired == ["pre", "post"] * 5, fired
    for i in range(2):
        p, q, r = pa[i].detach(), pb[i].detach(), pc[i].detach()
        assert float((p - q).abs().max()) <= 4e-6 * float(q.abs().max())
        assert float((p - r).abs().max()) > 1e-4 * float(r.abs().max())           # the two decay forms really differ on this trajectory
    assert float(oa.state[pa[0]]["step"]) == 5.0 and float(oa.state[pa[3]]["step"]) == 5.0


def test_failed_launch_leaves_step_counters_untouched(gpu_device):
    """A pag_adam_step call the library rejects (here: lr < 0 smuggled into the group after construction) raises and leaves the step
    counters where they were - no step is counted for an update that was not applied."""
    import pagnerf_amd
    dev = gpu_device
    p = torch.nn.Parameter(torch.randn(1000, device=dev))
    o = pagnerf_amd.optim.Adam([p], lr=1e-2, eps=1e-15)
    p.grad = torch.ones_like(p)
    o.step()
    o.param_groups[0]["lr"] = -1.0
    before = p.detach().clone()
    with pytest.raises(Exception):
        o.step()
    assert float(o.state[p]["step"]) == 1.0 and torch.equal(p.detach(), before)
    o.param_groups[0]["lr"] = 1e-2
    o.step()
    assert float(o.state[p]["step"]) == 2.0 and o._plans[0].count == 2
