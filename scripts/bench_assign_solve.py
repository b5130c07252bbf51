"""Device time of pag_assign_solve on assignment problems shaped like the late-training step's: B images x n labels x 199 columns.
usage: python scripts/bench_assign_solve.py   -> one line per case: us per launch (HIP events over 50 launches)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pagnerf_amd import ops, _lib as L

dev = torch.device("cuda:0")
rs = np.random.RandomState(0)
B, R, C = 6, 199, 199


def run(name, cost, n, lo_hi=None):
    info = torch.tensor([[n, 0]] * B, dtype=torch.int32, device=dev)
    d_cost = torch.from_numpy(cost.astype(np.float32)).to(dev)
    d_lh = torch.from_numpy(lo_hi.astype(np.int32)).to(dev) if lo_hi is not None else None
    targets = torch.zeros(B, R, dtype=torch.int64, device=dev)
    status = torch.zeros(B, dtype=torch.int32, device=dev)
    call = lambda: ops._call("pag_assign_solve", d_cost.data_ptr(), B, R, C, info.data_ptr(), d_lh.data_ptr() if d_lh is not None else None, targets.data_ptr(),
                             status.data_ptr(), L.stream())
    for _ in range(3):
        call()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        call()
    b.record()
    torch.cuda.synchronize()
    print("%-58s n = %3d  %8.1f us per launch   status %s" % (name, n, a.elapsed_time(b) / 50 * 1e3, status.cpu().tolist()))


def mat(kind, n):
    c = np.zeros((B, R, C))
    for b in range(B):
        if kind == "random":
            c[b, :n] = -rs.dirichlet(np.ones(C) * 0.5, size=n)
        else:      # an untrained head: every ray predicts about the same distribution, so every label's mean row is about the same
            c[b, :n] = -np.tile(rs.dirichlet(np.ones(C) * 0.05), (n, 1)) + rs.rand(n, C) * 1e-3
    return c


for n in (24, 60, 199):
    run("random rows (each label prefers its own columns)", mat("random", n), n)
    run("all rows alike (untrained head: long augmenting paths)", mat("alike", n), n)
lh = np.zeros((B, R, 2))
lh[..., 0], lh[..., 1] = 50, 80
run("all rows alike + the same id range [50, 80] for every label", mat("alike", 24), 24, lh)
