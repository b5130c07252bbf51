"""Soak of the round-6 paths together: the late-training step (voxel march, all channels, LinAssignmentThingsLoss on the device solver, segment regulariser;
alternating ONE-backward and two-call steps, the second with the assignment on its side stream) with fresh rays every step, and every 100 steps a validation
render of 64 000 rays through batch_render (the march two packs ahead on the high-priority stream) checked against the plain per-pack loop bit for bit.
   python scripts/soak_late_step.py [steps]   -> losses finite, device-solver status clean, allocator high-water mark flat, renders identical."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import pagnerf_amd
from pagnerf_amd.loss import LinAssignmentThingsLoss

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
args = bench.parse(["--raymarch", "voxel", "--samples", "2"])
dev = torch.device("cuda:0")
nef = bench.make_model(args, dev, seed=0)
bench.synthetic_prune(nef, 0.1)
nef.train()
tr_one, tr_two = bench.make_tracer(args), bench.make_tracer(args)
tr_two.graph_split = True
opt = bench.make_optimizer(nef)
fn = LinAssignmentThingsLoss()
channels = {"rgb", "depth", "semantics", "inst_embedding"}
images = 4
val_rays, _ = bench.make_rays(64000, dev, seed=5)
val_tracer = pagnerf_amd.PanopticPackedRFTracer(raymarch_type="voxel", num_steps=2, bg_color="white", ray_max_travel=6.0, use_graphs=False)
pipe = pagnerf_amd.Pipeline(nef, val_tracer)
marks, renders = [], 0
torch.cuda.synchronize()
t0 = time.perf_counter()
for it in range(steps):
    rays, gt = bench.make_rays(4096, dev, seed=2000 + it)
    two = it % 2 == 1
    loss = bench.train_step(nef, tr_two if two else tr_one, opt, rays, gt, channels, 1, lin_assign=fn, images=images, seg_reg=True, overlap=two)
    if it % 100 == 99:
        nef.eval()
        with torch.no_grad():
            a = pagnerf_amd.batch_render(pipe, val_rays, channels=sorted(channels), render_batch=8000)
            b = [pipe(rays=r, lod_idx=None, channels=sorted(channels)) for r in val_rays.split(8000)]
        for ch in ("rgb", "depth", "semantics", "inst_embedding", "alpha"):
            assert torch.equal(getattr(a, ch), torch.cat([getattr(p, ch) for p in b], 0)), (it, ch)
        renders += 1
        nef.train()
    if it % max(1, steps // 10) == 0 or it == steps - 1:
        torch.cuda.synchronize()
        assert torch.isfinite(loss), (it, loss)
        marks.append((it, round(float(loss), 3), torch.cuda.memory_allocated() >> 20, torch.cuda.max_memory_allocated() >> 20, fn.solver))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
for m in marks:
    print("step %5d  loss %10.3f  allocated %6d MiB  peak %6d MiB  solver %s" % m)
print("%d steps + %d validation renders in %.2f s" % (steps, renders, dt))
assert fn.solver == "device", "the device assignment reported a failure and the loss fell back to the host solver"
assert marks[-1][3] - marks[3][3] < 128, "allocator peak keeps growing"
assert marks[-1][1] < marks[0][1], "the objective did not go down"
print("soak ok")
