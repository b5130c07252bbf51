"""Linear-assignment losses with the device-side cost matrix against the golden vectors produced by the
reference's loss/lin_assignment.py and loss/lin_assignment_things.py (tests/golden/g5_linassign.npz): virtual labels
bit-exact, loss values to fp32 tolerance; pag_label_sums itself against a numpy sequential reduction."""
import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu


def test_label_sums_kernel(gpu_device):
    from pagnerf_amd import loss as pl
    dev = gpu_device
    rs = np.random.RandomState(3)
    for P, C, col0 in ((1, 5, 0), (255, 200, 1), (4096, 200, 1), (5000, 3, 0), (777, 600, 7)):
        vals = rs.uniform(size=(P, C)).astype(np.float32)
        gt = rs.choice([-1, 0, 2, 5, 9, 1000], size=P).astype(np.int64)
        mask = rs.uniform(size=P) > 0.3
        labels = [0, 2, 5, 9, 77, 1000]
        for use_mask in (False, True):
            s, c = pl.label_sums(torch.from_numpy(vals).to(dev), torch.from_numpy(gt).to(dev), labels, col0=col0,
                                 row_mask=torch.from_numpy(mask).to(dev) if use_mask else None)
            for k, lab in enumerate(labels):
                sel = (gt == lab) & (mask if use_mask else True)
                ref = np.zeros(C - col0, dtype=np.float32)
                for row in vals[sel]:                                   # sequential fp32 sum in ray order
                    ref = (ref + row[col0:]).astype(np.float32)
                assert int(c[k]) == int(sel.sum())
                assert np.array_equal(s[k].cpu().numpy(), ref), (P, C, lab)
        sb, _ = pl.label_sums(torch.from_numpy(vals).to(dev).bfloat16(), torch.from_numpy(gt).to(dev), labels, col0=col0)
        ref = np.stack([torch.from_numpy(vals).bfloat16().float().numpy()[gt == lab][:, col0:].sum(0) for lab in labels])
        np.testing.assert_allclose(sb.cpu().numpy(), ref, rtol=1e-4, atol=1e-4)


def test_g5_virtual_labels_and_losses(gpu_device):
    from pagnerf_amd import loss as pl
    dev = gpu_device
    g = golden("g5_linassign.npz")
    prob = torch.from_numpy(g["prob"]).to(dev)
    logits = torch.from_numpy(g["logits"]).to(dev)
    gt = torch.from_numpy(g["gt"]).to(dev)
    stuff = torch.from_numpy(g["stuff"]).to(dev)
    pts = torch.from_numpy(g["points_3d"]).to(dev)
    plain = pl.LinAssignmentLoss()
    for b in range(prob.shape[0]):
        v = plain.create_virtual_gt_with_linear_assignment(gt[b], logits[b])
        assert np.array_equal(v.cpu().numpy(), g["virt_plain"][b])
    np.testing.assert_allclose(plain(prob, gt).cpu().numpy(), g["loss_plain"], rtol=1e-5)
    for tag, rej in (("things", False), ("things_rej", True)):
        lo = pl.LinAssignmentThingsLoss(outlier_rejection=rej)
        for b in range(prob.shape[0]):
            vm = (stuff[b] | (gt[b] > 0))
            v = lo.create_virtual_gt_with_linear_assignment(prob[b], torch.where(vm, gt[b], torch.zeros_like(gt[b])),
                                                            pts[b] if rej else None)
            assert np.array_equal(v[vm].cpu().numpy(), g[f"virt_{tag}_{b}"]), tag
        p = prob.clone().requires_grad_(True)
        out = lo(p, gt, stuff, pts if rej else None)
        np.testing.assert_allclose(out.detach().cpu().numpy(), g[f"loss_{tag}"], rtol=1e-5, atol=1e-6)
        out.sum().backward()
        assert torch.isfinite(p.grad).all() and float(p.grad.abs().sum()) > 0


def test_one_sync_path_virtual_labels_bit_exact_vs_reference(gpu_device):
    """north_star: instance-ID linear-assignment indices bit-exact.  The DEFAULT path (pag_assign_cost -> SciPy -> pag_assign_nll_fwd, both images of the
    golden batch in one launch set) hands out the virtual labels it trained against; on the valid rays (stuff | id > 0, lin_assignment_things.py:60-62) they
    equal the labels the reference's own create_virtual_gt_with_linear_assignment produced (g5 virt_things_{b} / virt_things_rej_{b}), bit for bit."""
    from pagnerf_amd import loss as pl
    dev = gpu_device
    g = golden("g5_linassign.npz")
    prob = torch.from_numpy(g["prob"]).to(dev)
    gt = torch.from_numpy(g["gt"]).to(dev)
    stuff = torch.from_numpy(g["stuff"]).to(dev)
    pts = torch.from_numpy(g["points_3d"]).to(dev)
    assert prob.shape[0] == 2
    for tag, rej in (("things", False), ("things_rej", True)):
        lo = pl.LinAssignmentThingsLoss(outlier_rejection=rej)
        out = lo(prob.clone().requires_grad_(True), gt, stuff, pts if rej else None)
        virt = lo.last_virtual_labels
        assert virt is not None and virt.dtype == torch.int64 and virt.shape == gt.shape and not virt.requires_grad, "the fast path did not run"
        for b in range(2):
            vm = (stuff[b] | (gt[b] > 0))
            assert np.array_equal(virt[b][vm].cpu().numpy(), g[f"virt_{tag}_{b}"]), (tag, b)
        np.testing.assert_allclose(out.detach().cpu().numpy(), g[f"loss_{tag}"], rtol=1e-5, atol=1e-6)


def test_one_sync_assignment_path_equals_general_path(gpu_device):
    """LinAssignmentThingsLoss: the one-synchronisation path (pag_assign_cost + _AssignNLL, ABI 10) against the general path (explicit label list, tensor ops) -
    loss values and gradients on the golden batch and on random batches with stuff-only rays, invalid rays, an image without things, an image whose
    prediction is already right (loss 0), and large arbitrary ids."""
    from pagnerf_amd import loss as pl
    dev = gpu_device
    g = golden("g5_linassign.npz")
    cases = [(torch.from_numpy(g["prob"]).to(dev), torch.from_numpy(g["gt"]).to(dev), torch.from_numpy(g["stuff"]).to(dev))]
    gen = torch.Generator().manual_seed(5)
    B, P, I = 3, 777, 40
    prob = torch.softmax(torch.randn(B, P, I, generator=gen) * 3, -1)
    gt = torch.randint(-1, 9, (B, P), generator=gen) * 3            # ids -3, 0, 3 .. 24
    gt[1] = 0                                                        # an image without things
    stuff = torch.rand(B, P, generator=gen) > 0.5
    # image 2: the prediction already matches what the assignment will pick (every things ray puts its mass on one column per id)
    ids = gt[2].clamp(min=0)
    onehot = torch.full((P, I), 1e-3)
    onehot[torch.arange(P), torch.where(ids > 0, ids // 3 + 4, torch.zeros_like(ids))] = 1.0
    prob[2] = onehot / onehot.sum(-1, keepdim=True)
    cases.append((prob.to(dev), gt.to(dev), stuff.to(dev)))
    big = gt.clone()
    big[0, :5] = 5000000000                                          # ids are arbitrary int64 values
    cases.append((prob.to(dev), big.to(dev), stuff.to(dev)))
    for n, (p, t, m) in enumerate(cases):
        fast, slow = pl.LinAssignmentThingsLoss(), pl.LinAssignmentThingsLoss()
        slow.fast_path = False
        pf, ps = p.clone().requires_grad_(True), p.clone().requires_grad_(True)
        lf, ls = fast(pf, t, m), slow(ps, t, m)
        assert lf.shape == ls.shape == p.shape[:2]
        np.testing.assert_allclose(lf.detach().cpu().numpy(), ls.detach().cpu().numpy(), rtol=2e-6, atol=1e-7, err_msg=str(n))
        assert np.array_equal((lf != 0).cpu().numpy(), (ls != 0).cpu().numpy())
        w = torch.rand(lf.shape, device=dev)
        (lf * w).sum().backward()
        (ls * w).sum().backward()
        np.testing.assert_allclose(pf.grad.cpu().numpy(), ps.grad.cpu().numpy(), rtol=2e-6, atol=1e-7, err_msg=str(n))
        if n == 1:      # image 1 has no things: its stuff rays are trained towards column 0; image 2's prediction is already the assignment: no loss
            assert float(lf[1].detach().abs().sum()) > 0 and float(lf[2].detach().abs().sum()) == 0 and float(lf[0].detach().abs().sum()) > 0
    # more distinct ids than the device-side set holds (1024): pag_assign_cost reports it and the module takes the general path by itself
    P2, I2 = 3000, 12
    many = (torch.arange(P2) % 1500 + 1)[None].to(dev)
    pm = torch.softmax(torch.randn(1, P2, I2, generator=gen), -1).to(dev)
    sm = torch.zeros(1, P2, dtype=torch.bool, device=dev)
    # (host solver: the device solver reports the same condition one call later - test_device_and_host_solver_paths_are_identical)
    fast, slow = pl.LinAssignmentThingsLoss(solver="scipy"), pl.LinAssignmentThingsLoss()
    slow.fast_path = False
    assert fast._fast(pm, many, sm.view(torch.uint8)) is None
    assert torch.equal(fast(pm, many, sm), slow(pm, many, sm))
    # one pending begin() per loss object: a second begin() before finish() would overwrite the first call's cost rows / targets / event -
    # it raises instead; after finish() the object is free again, and finish(begin()) equals forward()
    p, t, m = cases[1]
    two = pl.LinAssignmentThingsLoss()
    pend = two.begin(p, t, m)
    with pytest.raises(RuntimeError, match="before finish"):
        two.begin(p, t, m)
    first = two.finish(pend)
    assert torch.equal(first, two.finish(two.begin(p, t, m))) and torch.equal(first, pl.LinAssignmentThingsLoss()(p, t, m))
    # with outlier rejection (best.yaml:106 `inst_outlier_rejection: true`): the id-range mask travels in the same copy as the cost
    p, t, m = cases[0]
    pts = torch.from_numpy(g["points_3d"]).to(dev)
    fast, slow = pl.LinAssignmentThingsLoss(outlier_rejection=True), pl.LinAssignmentThingsLoss(outlier_rejection=True)
    slow.fast_path = False
    pf, ps = p.clone().requires_grad_(True), p.clone().requires_grad_(True)
    lf, ls = fast(pf, t, m, pts), slow(ps, t, m, pts)
    np.testing.assert_allclose(lf.detach().cpu().numpy(), ls.detach().cpu().numpy(), rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(lf.detach().cpu().numpy(), g["loss_things_rej"], rtol=1e-5, atol=1e-6)
    lf.sum().backward()
    ls.sum().backward()
    np.testing.assert_allclose(pf.grad.cpu().numpy(), ps.grad.cpu().numpy(), rtol=2e-6, atol=1e-7)
    with pytest.raises(AssertionError):
        fast(p, t, m)                                                 # :36-37: outlier rejection requires the 3-D points
    # the kernels' own outputs: labels, counts-derived cost and virtual labels against the general path's pieces
    p, t, m = cases[0]
    lo = pl.LinAssignmentThingsLoss()
    for b in range(p.shape[0]):
        vm = m[b] | (t[b] > 0)
        gt_v = torch.where(vm, t[b], torch.zeros_like(t[b]))
        labels = sorted(torch.unique(gt_v[gt_v > 0]).cpu().tolist())[:p.shape[2] - 1]
        ref_cost = pl.cost_matrix(p[b], gt_v, labels, col0=1)
        w = lo._workspace(1, p.shape[1], p.shape[2], dev)
        from pagnerf_amd import _lib as L, ops
        ops._call("pag_assign_cost", p[b].data_ptr(), 1, p.shape[1], 0, p.stride(1), p.shape[2], 1, t[b].contiguous().data_ptr(), p.shape[2] - 1,
                  w["sums"].data_ptr(), w["counts"].data_ptr(), w["info"].data_ptr(), w["labels"].data_ptr(), w["cost"].data_ptr(), None, 0.0, 0.0, 0, None, None, None,
                  L.stream())
        torch.cuda.synchronize()
        n = int(w["info"][0, 0])
        assert n == len(labels) and int(w["info"][0, 1]) == 0 and w["labels"][0, :n].cpu().tolist() == labels
        assert np.array_equal(w["cost"][0, :n].cpu().numpy().astype(np.float64), ref_cost)


def test_render_loss_matches_trainer_formulas(gpu_device):
    """pag_render_loss_fwd / _bwd against the reference trainer's arithmetic written as plain tensor ops in float64
    (pc_nerf/trainer.py:443-446 rgb L1; :459-465 semantics: nll 'none' / temperature * conf, mean over all rays;
    loss/lin_assignment_things.py:80 instance NLL; F.nll_loss 'mean' = mean over the rows whose target is not ignored).
    Tolerance: fp32 sums of <= 1e5 terms against float64 - 2e-6 relative on the loss, 1e-6 relative on gradients."""
    import torch.nn.functional as F
    from pagnerf_amd import loss as pl
    dev = gpu_device
    g = torch.Generator().manual_seed(5)
    for N, Cs, Ci in ((1, 3, 4), (300, 7, 200), (4096, 7, 200), (100000, 5, 33)):
        rgb = torch.rand(N, 3, generator=g)
        gt = torch.rand(N, 3, generator=g)
        gt[::7] = rgb[::7]                                                   # exact zeros: sgn(0) = 0
        sem = torch.softmax(torch.randn(N, Cs, generator=g) * 3, -1)
        inst = torch.softmax(torch.randn(N, Ci, generator=g) * 6, -1)
        sem_t = torch.randint(0, Cs, (N,), generator=g)
        inst_t = torch.randint(0, Ci, (N,), generator=g)
        inst_t[::5] = -100                                                   # ignore_index rows
        conf = torch.rand(N, generator=g)
        for use_conf, temp, sem_mode in ((False, 1.0, "valid"), (True, 0.7, "all")):
            leaves = [t.clone().to(dev).requires_grad_(True) for t in (rgb, sem, inst)]
            loss, terms = pl.render_loss(leaves[0], gt.to(dev), 10.0,
                                         pl.NllTerm(leaves[1], sem_t.to(dev), weight=0.1, temperature=temp,
                                                    conf=conf.to(dev) if use_conf else None, mean_over=sem_mode),
                                         pl.NllTerm(leaves[2], inst_t.to(dev), weight=1000.0))
            (loss * 0.5).backward()                                          # upstream gradient read on the device
            ref_leaves = [t.clone().double().requires_grad_(True) for t in (rgb, sem, inst)]
            r_rgb = 10.0 * torch.abs(ref_leaves[0] - gt.double()).mean()
            nll = F.nll_loss(torch.log(ref_leaves[1] + 1e-27) / temp, sem_t, reduction="none")
            if use_conf:
                nll = nll * conf.double()
            r_sem = 0.1 * (nll.mean() if sem_mode == "all" else nll.sum() / N)          # no ignored rows here: same thing
            r_inst = 1000.0 * F.nll_loss(torch.log(ref_leaves[2] + 1e-27), inst_t, reduction="mean")
            ref = r_rgb + r_sem + r_inst
            (ref * 0.5).backward()
            t = terms.cpu().double()
            for got, want in ((loss.item(), ref.item()), (t[1].item(), r_rgb.item()), (t[2].item(), r_sem.item()), (t[3].item(), r_inst.item())):
                if want != want:                 # N = 1: the only row is ignored -> 0/0 = NaN, as F.nll_loss gives
                    assert got != got
                    continue
                assert abs(got - want) <= 2e-6 * abs(want) + 1e-9, (N, got, want)
            assert t[5].item() == float((inst_t >= 0).sum())
            for a, b in zip(leaves, ref_leaves):
                ga, gb = a.grad.cpu().double(), b.grad
                assert torch.equal(ga == 0, gb == 0)
                assert torch.allclose(ga, gb, rtol=1e-5, atol=0.0, equal_nan=True), (N, (ga - gb).abs().max())
    # determinism (fixed-order partials): same bits on a second call; single terms work alone
    a = pl.render_loss(rgb.to(dev), gt.to(dev), 10.0, pl.NllTerm(sem.to(dev), sem_t.to(dev)), None)[0]
    b = pl.render_loss(rgb.to(dev), gt.to(dev), 10.0, pl.NllTerm(sem.to(dev), sem_t.to(dev)), None)[0]
    assert torch.equal(a, b)
    only = pl.render_loss(term_a=pl.NllTerm(inst.to(dev), inst_t.to(dev)))[0]
    assert abs(only.item() - F.nll_loss(torch.log(inst.double() + 1e-27), inst_t).item()) < 1e-5 * abs(only.item())


def test_segment_consistency_regularizer_vs_reference_golden(gpu_device):
    """pagnerf_amd.loss.segment_consistency_regularizer (device tensors, no host synchronisation) against the value and the autograd gradient the
    reference's loss/regularizers.py:5-35 produced for the g6 batch (skipped segment, forced-0 segment, one-ray segment, large ids, three images with
    different segment counts), and against the CPU oracle on a random batch."""
    from pagnerf_amd import loss as pl
    from oracle import regularizers as oreg
    dev = gpu_device
    g = golden("g6_reg.npz")
    x = (torch.from_numpy(g["seg_prob"]).to(dev) + 1e-27).requires_grad_(True)
    lab = torch.from_numpy(g["seg_labels"]).to(dev)
    val = pl.segment_consistency_regularizer(x, lab)
    val.backward()
    np.testing.assert_allclose(float(val.detach()), float(g["seg_reg"]), rtol=1e-5)
    assert np.array_equal(x.grad.cpu().numpy() != 0, g["seg_reg_grad"] != 0)
    np.testing.assert_allclose(x.grad.cpu().numpy(), g["seg_reg_grad"], rtol=1e-5, atol=0)
    rs = np.random.RandomState(4)
    B, P, I = 6, 4096, 200                                           # the shape of a best.yaml step
    prob = torch.softmax(torch.from_numpy(rs.standard_normal(size=(B, P, I)).astype(np.float32)) * 3, -1)
    labels = torch.from_numpy(rs.choice([0, 0, 0, 1001, 1002, 1007, 1013, 2000], size=(B, P)).astype(np.int64))
    want = oreg.segment_consistency_regularizer(prob.numpy() + np.float32(1e-27), labels.numpy())
    got = pl.segment_consistency_regularizer(prob.to(dev) + 1e-27, labels.to(dev))
    np.testing.assert_allclose(float(got), float(want), rtol=1e-5)
    # more distinct ids than slots: NaN, not a wrong number
    many = torch.arange(pl.SEGMENT_SLOTS + 5, device=dev)[None]
    pm = torch.softmax(torch.randn(1, pl.SEGMENT_SLOTS + 5, 8, device=dev), -1)
    assert torch.isnan(pl.segment_consistency_regularizer(pm, many))


def test_segment_regulariser_kernels_vs_tensor_ops_and_oracle(gpu_device):
    """pag_segment_reg_fwd / _bwd (ABI 12, the default on fp32 CUDA tensors) against the tensor-op form of the same function and the numpy oracle (value AND
    gradient) on the shape of a best.yaml step: ids of every sign and size, a segment that is skipped (all its rays predict column 0), a segment forced to
    label 0 (:29-30), a one-ray segment, images with different segment counts, `eps` added inside the kernels, a strided view as input, bitwise
    reproducibility, and more ids than the set holds."""
    from pagnerf_amd import loss as pl
    from oracle import regularizers as oreg
    dev = gpu_device
    rs = np.random.RandomState(9)
    B, P, I = 6, 4096, 200
    prob = torch.softmax(torch.from_numpy(rs.standard_normal(size=(B, P, I)).astype(np.float32)) * 3, -1)
    ids = np.array([0, 0, 0, -7, 3, 1001, 1002, 1007, 1013, 2000, 1 << 40, -(1 << 50)], dtype=np.int64)
    labels = torch.from_numpy(ids[rs.randint(0, len(ids), size=(B, P))])
    labels[1] = torch.from_numpy(rs.randint(0, 700, size=P).astype(np.int64))                 # an image with ~700 segments
    labels[2, :] = 5                                                                          # one segment
    labels[3, 17] = 999999                                                                    # a one-ray segment
    seg_skip = labels[0] == 1001                                                              # all its rays predict column 0: skipped (:24-25)
    prob[0, seg_skip] = torch.softmax(torch.cat([torch.full((int(seg_skip.sum()), 1), 9.0), torch.zeros(int(seg_skip.sum()), I - 1)], 1), -1)
    seg_zero = labels[0] == 1002                                                              # mostly column 0, a few column 3: label forced to 0 (:29-30)
    n0 = int(seg_zero.sum())
    forced = torch.zeros(n0, I)
    forced[:, 0] = 9.0
    forced[: max(1, n0 // 10), 3] = 12.0
    prob[0, seg_zero] = torch.softmax(forced, -1)
    want, want_grad = oreg.segment_consistency_regularizer(prob.numpy() + np.float32(1e-27), labels.numpy(), want_grad=True)

    def run(kernels, x, eps):
        pl.SEGMENT_KERNELS = kernels
        try:
            x = x.clone().requires_grad_(True)
            v = pl.segment_consistency_regularizer(x, labels.to(dev), eps=eps)
            assert (type(v.grad_fn).__name__ == "_SegmentRegBackward") == kernels
            (v * 3.0).backward()
            return v.detach(), x.grad / 3.0
        finally:
            pl.SEGMENT_KERNELS = True
    vk, gk = run(True, prob.to(dev), 1e-27)
    vt, gt = run(False, prob.to(dev), 1e-27)
    np.testing.assert_allclose(float(vk), float(want), rtol=2e-6)
    np.testing.assert_allclose(float(vk), float(vt), rtol=2e-6)
    assert np.array_equal(gk.cpu().numpy() != 0, want_grad != 0)
    np.testing.assert_allclose(gk.cpu().numpy(), want_grad, rtol=2e-5, atol=0)
    np.testing.assert_allclose(gk.cpu().numpy(), gt.cpu().numpy(), rtol=2e-5, atol=0)
    vk2, gk2 = run(True, prob.to(dev), 1e-27)
    assert torch.equal(vk, vk2) and torch.equal(gk, gk2)                                       # bitwise reproducible
    # a strided view (row stride 256) gives the same bits as the contiguous tensor
    wide = torch.zeros(B, P, 256, device=dev)
    wide[..., :I] = prob.to(dev)
    vs, gs = run(True, wide[..., :I], 1e-27)
    assert torch.equal(vs, vk) and torch.equal(gs, gk)
    # eps outside (the reference's call) == eps inside
    vo, go = run(True, prob.to(dev) + 1e-27, 0.0)
    assert torch.equal(vo, vk) and torch.equal(go, gk)
    # more distinct ids than slots: NaN and a zero gradient, not a wrong number
    many = torch.arange(pl.SEGMENT_SLOTS + 5, device=dev)[None]
    pm = torch.softmax(torch.randn(1, pl.SEGMENT_SLOTS + 5, 8, device=dev), -1).requires_grad_(True)
    v = pl.segment_consistency_regularizer(pm, many)
    assert torch.isnan(v)
    v.backward()
    assert float(pm.grad.abs().sum()) == 0.0


def test_rays_to_3d_points_vs_oracle(gpu_device):
    """BAPipeline.rays_to_3d_points / _indexed (utils/outlier_rejection.py:74-97 through the restated camera transform) against
    oracle.regularizers.rays_to_3d_points per camera."""
    from pagnerf_amd.ba_pipeline import BAPipeline, rotation_6d_to_matrix
    from oracle import regularizers as oreg
    import pagnerf_amd
    dev = gpu_device
    gen = torch.Generator().manual_seed(3)
    C, n = 3, 50
    views = torch.eye(4).repeat(C, 1, 1)
    q, _ = torch.linalg.qr(torch.randn(C, 3, 3, generator=gen))
    views[:, :3, :3] = q
    views[:, :3, 3] = torch.randn(C, 3, generator=gen) * 0.1
    pipe = BAPipeline(torch.nn.Module(), views).to(dev)
    o, d = torch.randn(C * n, 3, generator=gen) * 0.01, torch.nn.functional.normalize(torch.randn(C * n, 3, generator=gen), dim=-1)
    depth = torch.rand(C * n, 1, generator=gen)
    got = pipe.rays_to_3d_points(pagnerf_amd.Rays(o.to(dev), d.to(dev)), depth.to(dev), torch.arange(C)).cpu()
    got_i = pipe.rays_to_3d_points_indexed(o.to(dev), d.to(dev), depth.to(dev), torch.arange(C * n, device=dev) // n).cpu()
    R = rotation_6d_to_matrix(pipe.camera_extrinsics.detach().cpu()[:, :6])
    t = pipe.camera_extrinsics.detach().cpu()[:, 6:]
    for c in range(C):
        sl = slice(c * n, (c + 1) * n)
        want = oreg.rays_to_3d_points(o[sl].numpy(), d[sl].numpy(), depth[sl, 0].numpy(), R[c].numpy(), t[c].numpy())
        np.testing.assert_allclose(got[sl].numpy(), want, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(got_i[sl].numpy(), want, rtol=1e-5, atol=1e-6)


def test_segment_regulariser_kernel_edge_shapes(gpu_device):
    """pag_segment_reg_fwd / _bwd at the edges of their argument ranges, against the numpy oracle: one ray, one image, one column (every segment is
    skipped: no column 1.. exists - value 0, zero gradient), P not a multiple of the launch granules, the reserved id INT64_MIN (NaN, zero gradient)."""
    from pagnerf_amd import loss as pl
    from oracle import regularizers as oreg
    dev = gpu_device
    rs = np.random.RandomState(2)
    for B, P, I in ((1, 1, 5), (1, 7, 1), (3, 1001, 9), (2, 64, 2)):
        prob = torch.softmax(torch.from_numpy(rs.standard_normal(size=(B, P, I)).astype(np.float32)) * 2, -1)
        labels = torch.from_numpy(rs.randint(-3, 4, size=(B, P)).astype(np.int64))
        want, want_grad = oreg.segment_consistency_regularizer(prob.numpy() + np.float32(1e-27), labels.numpy(), want_grad=True)
        x = prob.to(dev).requires_grad_(True)
        v = pl.segment_consistency_regularizer(x, labels.to(dev), eps=1e-27)
        assert type(v.grad_fn).__name__ == "_SegmentRegBackward"
        v.backward()
        np.testing.assert_allclose(float(v.detach()), float(want), rtol=2e-6, atol=0, err_msg=str((B, P, I)))
        np.testing.assert_allclose(x.grad.cpu().numpy(), want_grad, rtol=2e-5, atol=0, err_msg=str((B, P, I)))
    lab = torch.zeros(1, 16, dtype=torch.int64, device=dev)
    lab[0, 3] = torch.iinfo(torch.int64).min
    x = torch.softmax(torch.randn(1, 16, 4, device=dev), -1).requires_grad_(True)
    v = pl.segment_consistency_regularizer(x, lab)
    assert torch.isnan(v)
    v.backward()
    assert float(x.grad.abs().sum()) == 0.0


def test_device_hungarian_equals_scipy_bit_exact(gpu_device):
    """pag_assign_solve (ABI 13: the Hungarian step of loss/lin_assignment_things.py:45 on the device, one wave per image) against
    scipy.optimize.linear_sum_assignment(np.nan_to_num(cost)) itself: the SAME assigned columns (north_star: indices bit-exact) on batches of images with
    different label counts - random fp32 costs, tie-heavy integer costs, constant matrices, NaN / inf entries, the outlier-rejection mask, an image without
    labels, one with as many labels as columns (no LDS staging), and the status codes (id-set overflow -> 1, all-ones targets)."""
    import scipy.optimize
    from pagnerf_amd import ops
    from pagnerf_amd import _lib as L
    dev = gpu_device
    rs = np.random.RandomState(3)
    for case in range(12):
        B = 6
        R = C = [199, 199, 24, 199, 64, 256, 199, 31, 199, 2, 1, 199][case]
        n = rs.randint(0, min(R, 40) + 1, size=B)
        n[0] = 0
        if case in (1, 5):
            n[1] = R                                                   # as many labels as columns: the matrix does not fit the LDS stage
        cost = np.zeros((B, R, C), dtype=np.float32)
        for b in range(B):
            kind = (case + b) % 5
            if kind == 0:
                cost[b] = -rs.rand(R, C)
            elif kind == 1:
                cost[b] = rs.randint(0, 3, (R, C))
            elif kind == 2:
                cost[b] = 0.25
            elif kind == 3:
                cost[b] = rs.randn(R, C)
                cost[b][rs.rand(R, C) < 0.02] = np.nan
                cost[b][rs.rand(R, C) < 0.01] = np.inf
            else:
                cost[b] = -np.round(rs.rand(R, C) * 8) / 8
        use_mask = case % 2 == 1
        lo = rs.randint(0, max(C - 1, 1), size=(B, R))
        lo_hi = np.stack([lo, np.minimum(lo + rs.randint(0, 60, size=(B, R)), C - 1)], -1).astype(np.int32)
        info = np.stack([n, np.zeros(B, dtype=np.int64)], -1).astype(np.int32)
        if case == 3:
            info[2, 1] = 1                                             # pag_assign_cost's id set overflowed for image 2
        d_cost, d_info, d_lh = torch.from_numpy(cost).to(dev), torch.from_numpy(info).to(dev), torch.from_numpy(lo_hi).to(dev)
        targets = torch.full((B, R), -7, device=dev, dtype=torch.int64)
        status = torch.full((B,), -1, device=dev, dtype=torch.int32)
        ops._call("pag_assign_solve", d_cost.data_ptr(), B, R, C, d_info.data_ptr(), d_lh.data_ptr() if use_mask else None, targets.data_ptr(), status.data_ptr(), L.stream())
        torch.cuda.synchronize()
        got, st = targets.cpu().numpy(), status.cpu().numpy()
        for b in range(B):
            want = np.ones(R, dtype=np.int64)
            if info[b, 1]:
                assert st[b] == 1 and np.array_equal(got[b], want), (case, b)
                continue
            c64 = cost[b, :n[b]].astype(np.float64)
            if use_mask:
                ids = np.arange(C)[None, :]
                c64[~((lo_hi[b, :n[b], :1] <= ids) & (ids <= lo_hi[b, :n[b], 1:]))] = 10000
            rows, cols = scipy.optimize.linear_sum_assignment(np.nan_to_num(c64))
            want[rows] = cols + 1
            assert st[b] == 0 and np.array_equal(got[b], want), (case, b, int(n[b]), np.nonzero(got[b] != want)[0][:5])


def test_device_and_host_solver_paths_are_identical(gpu_device):
    """LinAssignmentThingsLoss(solver="device") - the default: no host wait - against solver="scipy" (one copy + wait, SciPy per image): the same virtual labels,
    loss values and gradients bit for bit, with and without outlier rejection, on the golden batch (whose labels the reference produced) and random batches;
    begin() / finish() behave alike; a batch whose ids overflow the device-side set is reported at the NEXT call and the object switches to the host solver."""
    import warnings
    from pagnerf_amd import loss as pl
    dev = gpu_device
    g = golden("g5_linassign.npz")
    p0, t0, m0 = torch.from_numpy(g["prob"]).to(dev), torch.from_numpy(g["gt"]).to(dev), torch.from_numpy(g["stuff"]).to(dev)
    pts0 = torch.from_numpy(g["points_3d"]).to(dev)
    gen = torch.Generator().manual_seed(11)
    B, P, I = 6, 4096, 200
    prob = torch.softmax(torch.randn(B, P, I, generator=gen) * 2, -1).to(dev)
    gt = (torch.randint(0, 30, (B, P), generator=gen) * (torch.rand(B, P, generator=gen) > 0.3)).to(dev)
    stuff = (torch.rand(B, P, generator=gen) > 0.5).to(dev)
    pts = (torch.rand(B, P, 3, generator=gen) * 2 - 1).to(dev)
    for (p, t, m, q) in ((p0, t0, m0, None), (p0, t0, m0, pts0), (prob, gt, stuff, None), (prob, gt, stuff, pts)):
        rej = q is not None
        dv, hs = pl.LinAssignmentThingsLoss(outlier_rejection=rej), pl.LinAssignmentThingsLoss(outlier_rejection=rej, solver="scipy")
        assert dv.solver == "device" and hs.solver == "scipy"
        pd_, ph = p.clone().requires_grad_(True), p.clone().requires_grad_(True)
        args = (t, m) if q is None else (t, m, q)
        ld, lh = dv(pd_, *args), hs(ph, *args)
        assert torch.equal(dv.last_virtual_labels, hs.last_virtual_labels) and torch.equal(ld, lh)
        w = torch.rand(ld.shape, device=dev)
        (ld * w).sum().backward()
        (lh * w).sum().backward()
        assert torch.equal(pd_.grad, ph.grad)
        assert torch.equal(dv.finish(dv.begin(p, *args)), ld.detach())
    # the golden labels through the device solver (the reference's own output)
    dv = pl.LinAssignmentThingsLoss()
    dv(p0, t0, m0)
    for b in range(p0.shape[0]):
        valid = (g["stuff"][b] | (g["gt"][b] > 0))
        assert np.array_equal(dv.last_virtual_labels[b].cpu().numpy()[valid], g["virt_things_%d" % b])       # (the fixture holds the valid rays)
    # more distinct ids than the device-side set holds: status 1, noticed at a later call, the object falls back to the host solver for good
    P2, I2 = 3000, 12
    many = (torch.arange(P2) % 1500 + 1)[None].to(dev)
    pm = torch.softmax(torch.randn(1, P2, I2, generator=gen), -1).to(dev)
    sm = torch.zeros(1, P2, dtype=torch.bool, device=dev)
    dv, hs = pl.LinAssignmentThingsLoss(), pl.LinAssignmentThingsLoss(solver="scipy")
    dv(pm, many, sm)
    torch.cuda.synchronize()
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        second = dv(pm, many, sm)
    assert dv.solver == "scipy" and any("switching to solver='scipy'" in str(c.message) for c in caught)
    assert torch.equal(second, hs(pm, many, sm))


def test_training_with_the_device_solver_follows_the_host_solver_bit_for_bit(gpu_device):
    """Twelve Adam steps of the late-training objective (rgb L1 + the per-image Hungarian instance term on the rendered probabilities, trainer.py:443-446,483-533)
    on a small scene, once with LinAssignmentThingsLoss(solver="device") and once with solver="scipy", from the same initial state: the assignment indices are
    the same integers every step, so the losses - and with them the parameters - are IDENTICAL, not merely close; ten of the steps also change the labels the
    assignment picks (the head is being trained), so this is the solver under moving cost matrices, not one matrix twelve times."""
    import copy
    import pagnerf_amd
    import test_gpu_parity as T
    from pagnerf_amd import loss as pl
    dev = gpu_device
    N, S, B = 512, 32, 2
    nef0, tracer, rays, occ, jitter = T._make_scene(dev, "bf16", N=N, S=S, cap_log2=12)
    gen = torch.Generator().manual_seed(21)
    gt_rgb = torch.rand(N, 3, generator=gen).to(dev)
    ids = (torch.arange(N) * 9 // N + 1).reshape(B, -1)                      # nine instances over the two images
    ids = torch.where(torch.rand(B, N // B, generator=gen) < 0.2, torch.zeros_like(ids), ids).to(dev)
    stuff = (ids == 0)
    jit = jitter.to(dev)
    curves, labels = {}, {}
    for solver in ("device", "scipy"):
        nef = copy.deepcopy(nef0)
        opt = pagnerf_amd.optim.Adam(nef.parameters(), lr=1e-2, eps=1e-15)
        fn = pl.LinAssignmentThingsLoss(solver=solver)
        out, lab = [], []
        for it in range(12):
            opt.zero_grad(set_to_none=True)
            rb = tracer(nef, channels={"rgb", "inst_embedding"}, rays=rays, jitter=jit, stage="train")
            inst = rb.inst_embedding.float().reshape(B, -1, rb.inst_embedding.shape[-1])
            loss = 10.0 * torch.abs(rb.rgb - gt_rgb).mean() + 100.0 * fn(inst, ids, stuff).mean()
            loss.backward()
            opt.step()
            out.append(loss.detach().clone())
            lab.append(fn.last_virtual_labels.clone())
        curves[solver], labels[solver] = torch.stack(out), torch.stack(lab)
    assert torch.equal(curves["device"], curves["scipy"]), (curves["device"] - curves["scipy"]).abs().max()
    assert torch.equal(labels["device"], labels["scipy"])
    assert float(curves["device"][-1]) < float(curves["device"][0])                                   # it trains
    assert int((labels["device"][1:] != labels["device"][:-1]).any(-1).any(-1).sum()) >= 1            # and the assignment moved while it did


def test_device_hungarian_equals_the_committed_scipy_answers(gpu_device):
    """pag_assign_solve on the 48 matrices of g10_lsap.npz (batched six at a time, padded to one shape): the committed SciPy 1.15.3 columns, bit for bit."""
    from pagnerf_amd import ops
    from pagnerf_amd import _lib as L
    dev = gpu_device
    g = golden("g10_lsap.npz")
    R = C = 64
    for b0 in range(0, 48, 6):
        cost = np.zeros((6, R, C), dtype=np.float32)
        info = np.zeros((6, 2), dtype=np.int32)
        want = np.ones((6, R), dtype=np.int64)
        for b in range(6):
            c = g["cost_%d" % (b0 + b)]
            # the kernel takes ONE column count per batch: pad the matrix with columns no row can prefer (they stay free, as if absent: +1e30 is never the minimum)
            cost[b, :c.shape[0], :c.shape[1]] = c
            cost[b, :c.shape[0], c.shape[1]:] = 1e30
            info[b, 0] = c.shape[0]
            want[b, :c.shape[0]] = g["cols_%d" % (b0 + b)] + 1
        d_cost, d_info = torch.from_numpy(cost).to(dev), torch.from_numpy(info).to(dev)
        targets = torch.zeros(6, R, device=dev, dtype=torch.int64)
        status = torch.zeros(6, device=dev, dtype=torch.int32)
        ops._call("pag_assign_solve", d_cost.data_ptr(), 6, R, C, d_info.data_ptr(), None, targets.data_ptr(), status.data_ptr(), L.stream())
        torch.cuda.synchronize()
        assert not status.any() and np.array_equal(targets.cpu().numpy(), want), b0
