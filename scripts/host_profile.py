"""Host-side (Python) time of the training step's head: wall-clock stamps from the ray-march read-back to the first encode
launch (the window in which the GPU idles), plus a cProfile of whole steps.  usage: python3 scripts/host_profile.py [steps]"""
import cProfile
import os
import pstats
import sys
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pagnerf_amd import ops

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
args = bench.parse(["--rays", "4096", "--samples", "512", "--grid", "permuto", "--precision", "bf16"])
dev = torch.device("cuda:0")
nef, tracer = bench.make_model(args, dev, 0), bench.make_tracer(args)
rays, gt = bench.make_rays(args.rays, dev, 1)
opt = bench.make_optimizer(nef)
chans = ["rgb", "semantics", "inst_embedding"]
for _ in range(5):
    bench.train_step(nef, tracer, opt, rays, gt, chans, 1)
torch.cuda.synchronize()

stamps = []
orig_call = ops._call


def stamped(name, *a):
    stamps.append((name, time.perf_counter()))
    return orig_call(name, *a)


ops._call = stamped
orig_march = ops.raymarch_ray


def march(*a, **k):
    r = orig_march(*a, **k)
    stamps.append(("march_return", time.perf_counter()))
    return r


ops.raymarch_ray = march
per = []
for _ in range(steps):
    stamps.clear()
    bench.train_step(nef, tracer, opt, rays, gt, chans, 1)
    names = [n for n, _ in stamps]
    t = dict()
    for n, ts in stamps:
        t.setdefault(n, ts)
    i_pack = names.index("march_return")
    per.append((stamps[i_pack + 1][1] - stamps[i_pack][1]) * 1e6)
    first = stamps[i_pack + 1][0]
torch.cuda.synchronize()
ops._call = orig_call
ops.raymarch_ray = orig_march
per.sort()
print("host us from the return of raymarch_ray (sample count known) to the next C-ABI launch (%s): median %.0f  min %.0f" % (first, per[len(per) // 2], per[0]))

# cProfile of that window only
pr = cProfile.Profile()


def march2(*a, **k):
    r = orig_march(*a, **k)
    pr.enable()
    return r


def call2(name, *a):
    if name.endswith("encode_fwd") or name.endswith("encode_fwd_add"):
        pr.disable()
    return orig_call(name, *a)


ops.raymarch_ray, ops._call = march2, call2
for _ in range(steps):
    bench.train_step(nef, tracer, opt, rays, gt, chans, 1)
torch.cuda.synchronize()
pr.disable()
ops.raymarch_ray, ops._call = orig_march, orig_call
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(40)
