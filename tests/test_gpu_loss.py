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
