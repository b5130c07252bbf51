"""Per-step device time of the headline step right after the bench's warm-up: is the K = 20 timed region slower than the sustained 300 steps because of its first
replays?  One event per step on the launch stream (no host sync inside the series), elapsed time between consecutive events.
usage: python scripts/step_series.py [bench flags]   ->  one line per phase with the per-step series in ms"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda:0")
args = bench.parse(sys.argv[1:])
nef, tracer = bench.make_model(args, dev, 0), bench.make_tracer(args, "ray")
rays, gt = bench.make_rays(args.rays, dev, 1000)
opt = bench.make_optimizer(nef)
chans = {"rgb", "depth", "semantics", "inst_embedding"}


def series(n):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(n):
        bench.train_step(nef, tracer, opt, rays, gt, chans, 1)
        ev[i + 1].record()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    return [ev[i].elapsed_time(ev[i + 1]) for i in range(n)], t_issue * 1e3 / n, wall * 1e3 / n


for _ in range(args.warmup):
    bench.train_step(nef, tracer, opt, rays, gt, chans, 1)
for tag, n in (("first 20 after warm-up", 20), ("next 20", 20), ("next 100", 100), ("after 2 s idle", 20)):
    if tag.startswith("after"):
        torch.cuda.synchronize()
        time.sleep(2.0)
    s, issue, wall = series(n)
    print("%-24s wall %.3f ms/step  host issue %.3f ms/step  device: mean %.3f  min %.3f  max %.3f  first5 %s  last5 %s" %
          (tag, wall, issue, sum(s) / n, min(s), max(s), " ".join("%.2f" % v for v in s[:5]), " ".join("%.2f" % v for v in s[-5:])), flush=True)
