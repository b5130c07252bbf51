"""Does the forward graph start right behind the ray march?  For the default bench workload: stream time from just after the march's last launch to the
end of the forward graph's last kernel inside real training steps, against the same graph replayed back to back (no march, no host work between)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pagnerf_amd import graphs

dev = torch.device("cuda:0")
args = bench.parse(sys.argv[1:])
nef, tracer = bench.make_model(args, dev, 0), bench.make_tracer(args, "ray")
rays, gt = bench.make_rays(args.rays, dev, 1)
opt = bench.make_optimizer(nef)
chans = {"rgb", "depth", "semantics", "inst_embedding"}
for _ in range(8):
    bench.train_step(nef, tracer, opt, rays, gt, chans, 1)
torch.cuda.synchronize()
orig = graphs._Graphed.__call__
pairs, last = [], []


def timed(self, *a, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = orig(self, *a, **k)
    e1.record()
    pairs.append((e0, e1))
    last[:] = [self]
    return out


graphs._Graphed.__call__ = timed
for _ in range(30):
    bench.train_step(nef, tracer, opt, rays, gt, chans, 1)
torch.cuda.synchronize()
in_step = sorted(a.elapsed_time(b) for a, b in pairs)
graphs._Graphed.__call__ = orig
g = last[0]
with torch.no_grad():
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        g.fwd.replay()
    e1.record()
    torch.cuda.synchronize()
print("forward graph inside a step (march's last launch -> graph end): median %.3f ms (min %.3f max %.3f); replayed back to back: %.3f ms each"
      % (in_step[len(in_step) // 2], in_step[0], in_step[-1], e0.elapsed_time(e1) / 20))
