"""Where the linear-assignment instance term's time goes (the late-training step, bench.py `with_lin_assignment`): every section of
pagnerf_amd.loss.LinAssignmentThingsLoss.forward on one rendered batch, each bracketed by device synchronisations (serialised cost: host + device),
and the whole term unbracketed."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.optimize
import torch
import bench
from pagnerf_amd import loss as Lm

dev = torch.device("cuda:0")
args = bench.parse(sys.argv[1:])
nef, tracer = bench.make_model(args, dev, 0), bench.make_tracer(args, "ray")
rays, gt = bench.make_rays(args.rays, dev, 1000)
chans = {"rgb", "depth", "semantics", "inst_embedding"}
opt = bench.make_optimizer(nef)
mod = Lm.LinAssignmentThingsLoss()
for _ in range(6):
    bench.train_step(nef, tracer, opt, rays, gt, chans, 1, lin_assign=mod)
torch.cuda.synchronize()


def sync_time(fn, n=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6, out


rb = tracer(nef, channels=chans, rays=rays, stage="train")
p = rb.inst_embedding.float().reshape(-1, rb.inst_embedding.shape[-1]).detach().requires_grad_(True)
g, m = gt["inst_ids"], gt["stuff"]
rows = []
t, valid = sync_time(lambda: torch.logical_or(m, g > 0)); rows.append(("valid mask", t))
t, gt_v = sync_time(lambda: torch.where(valid, g, torch.zeros_like(g))); rows.append(("masked gt", t))
things = gt_v > 0
t, labels = sync_time(lambda: sorted(torch.unique(gt_v[things]).cpu().tolist())[:p.shape[-1] - 1]); rows.append(("unique labels -> host", t))
t, cost = sync_time(lambda: Lm.cost_matrix(p, gt_v, labels, col0=1)); rows.append(("label sums + cost -> host", t))
t, rc = sync_time(lambda: scipy.optimize.linear_sum_assignment(np.nan_to_num(cost))); rows.append(("SciPy Hungarian %s" % (cost.shape,), t))
t, new = sync_time(lambda: Lm._lookup(gt_v, [labels[r] for r in rc[0]], [int(c) + 1 for c in rc[1]], 1)); rows.append(("relabel (table -> device, lookup)", t))
t, virt = sync_time(lambda: torch.where(things, new, torch.zeros_like(gt_v))); rows.append(("virtual labels", t))
t, wrong = sync_time(lambda: ((virt != p.argmax(dim=-1)) & valid).any()); rows.append(("argmax / any", t))


def nll_fwd():
    nll = -torch.log(p.gather(1, virt[:, None])[:, 0] + 1e-27)
    return torch.where(valid & wrong, nll, torch.zeros_like(nll))


t, out = sync_time(nll_fwd); rows.append(("gather / log / where", t))


def nll_bwd():
    p.grad = None
    (1000.0 * nll_fwd().mean()).backward()
    return p.grad


t, _ = sync_time(nll_bwd); rows.append(("the same + mean + backward to the probabilities", t))
for name, t in rows:
    print("%-52s %8.1f us" % (name, t))
tot = sum(t for _, t in rows[:-1])
print("sections (serialised, forward only): %.0f us" % tot)
t, _ = sync_time(lambda: mod(p[None], g[None], m[None]).mean())
print("LinAssignmentThingsLoss.forward + mean, one sync at the end: %.0f us" % t)
for la in (None, mod):
    for _ in range(3):
        bench.train_step(nef, tracer, opt, rays, gt, chans, 1, lin_assign=la)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        bench.train_step(nef, tracer, opt, rays, gt, chans, 1, lin_assign=la)
    torch.cuda.synchronize()
    print("train step %s: %.3f ms" % ("with the assignment term" if la is not None else "with a fixed-target NLL", (time.perf_counter() - t0) / 20 * 1e3))
