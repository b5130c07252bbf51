"""Soak of the graph / static-buffer paths with a sample count that changes every step: post-prune voxel march, fresh random rays per step.
   python scripts/soak_voxel.py [steps] [on|static]   -> captures / replays / overflows, step time, allocator high-water marks (must stay flat)."""
import os
import sys
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
mode = sys.argv[2] if len(sys.argv) > 2 else "on"
args = bench.parse(["--raymarch", "voxel", "--samples", "2", "--graphs", mode])
dev = torch.device("cuda:0")
nef = bench.make_model(args, dev, seed=0)
bench.synthetic_prune(nef, 0.1)
nef.train()
tracer = bench.make_tracer(args)
opt = bench.make_optimizer(nef)
channels = {"rgb", "depth", "semantics", "inst_embedding"}
counts, marks = [], []
torch.cuda.synchronize()
t0 = time.perf_counter()
for it in range(steps):
    rays, gt = bench.make_rays(4096, dev, seed=1000 + it)          # new origins / directions: the sample count moves by a few percent per step
    loss = bench.train_step(nef, tracer, opt, rays, gt, channels, 1)
    if it % max(1, steps // 10) == 0 or it == steps - 1:
        torch.cuda.synchronize()
        g = tracer._graphs
        marks.append((it, round(float(loss), 3), torch.cuda.memory_allocated() >> 20, torch.cuda.max_memory_allocated() >> 20,
                      (g.captures, g.replays, g.overflows) if g is not None else None))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
for m in marks:
    print("step %5d  loss %9.3f  allocated %6d MiB  peak %6d MiB  (captures, replays, overflows) %s" % m)
print("%d steps in %.2f s = %.3f ms per step (ray generation on the host included)" % (steps, dt, dt / steps * 1e3))
g = tracer._graphs
assert g is not None and g.replays > 0.9 * steps and g.captures <= 6, (g.captures, g.replays, g.overflows)
assert marks[-1][3] - marks[2][3] < 64, "allocator peak keeps growing"
print("soak ok")
