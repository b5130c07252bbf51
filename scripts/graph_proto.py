"""Feasibility probe: capture the post-march part of a train step (nef + compositing + loss + backward + Adam) in a HIP graph."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pagnerf_amd import ops

mode = sys.argv[1] if len(sys.argv) > 1 else "cfg0"
dev = torch.device("cuda:0")
if mode == "cfg0":
    args = bench.parse(["--rays", "256", "--samples", "64", "--grid", "hash"])
    nef, tracer = bench.make_model(args, dev, 0, grid="hash"), bench.make_tracer(args, "ray", 64)
    chans, n = {"rgb"}, 256
elif mode == "voxel":
    args = bench.parse(["--rays", "4096", "--raymarch", "voxel"])
    nef, tracer = bench.make_model(args, dev, 0), bench.make_tracer(args, "voxel")
    bench.synthetic_prune(nef, 0.1)
    chans, n = {"rgb", "depth", "semantics", "inst_embedding"}, 4096
else:
    args = bench.parse([])
    nef, tracer = bench.make_model(args, dev, 0), bench.make_tracer(args)
    chans, n = {"rgb", "depth", "semantics", "inst_embedding"}, 4096
rays, gt = bench.make_rays(n, dev, 1)
grid_params = [p for nm, p in nef.named_parameters() if "grid" in nm]
rest = [p for nm, p in nef.named_parameters() if "grid" not in nm]
opt = torch.optim.Adam([dict(params=grid_params, lr=1e-1), dict(params=rest, lr=1e-3)], eps=1e-15, fused=True, capturable=True)

# freeze the march: one real march, then every call returns the same tensors
g = nef.grid
real = g.raymarch(rays, level=None, num_samples=tracer.num_steps, raymarch_type=tracer.raymarch_type,
                  **({"max_travel": tracer.ray_max_travel} if tracer.raymarch_type == "voxel" else {}))
cache = g._pack_cache
def frozen(*a, **k):
    g._pack_cache = cache
    return real
g.raymarch = frozen

def step():
    opt.zero_grad(set_to_none=True)
    return bench.train_step.__wrapped__(nef, tracer, opt, rays, gt, chans, 1) if hasattr(bench.train_step, "__wrapped__") else _step()

def _step():
    from pagnerf_amd.loss import render_loss, NllTerm
    rb = tracer(nef, channels=chans, rays=rays, stage="train")
    if "semantics" in chans:
        loss, _ = render_loss(rb.rgb, gt["rgb"], 10.0, NllTerm(rb.semantics, gt["sem"], weight=0.1), NllTerm(rb.inst_embedding, gt["inst"], weight=1000.0))
    else:
        loss, _ = render_loss(rb.rgb, gt["rgb"], 10.0)
    loss.backward()
    opt.step()
    return loss

def eager(nsteps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(nsteps):
        opt.zero_grad(set_to_none=True)
        _step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / nsteps * 1e3

s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        _step()
torch.cuda.current_stream().wait_stream(s)
print("eager (march frozen): %.3f ms per step" % eager(100))
graph = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
with torch.cuda.graph(graph):
    static_loss = _step()
torch.cuda.synchronize()
l0 = float(static_loss)
t0 = time.perf_counter()
for _ in range(200):
    graph.replay()
torch.cuda.synchronize()
print("graph replay: %.3f ms per step, loss %.6f -> %.6f" % ((time.perf_counter() - t0) / 200 * 1e3, l0, float(static_loss)))
