"""Host-side profile of the launch-bound regimes: python3 scripts/host_profile_small.py voxel|voxelrgb|cfg0 [steps]
voxel: post-prune step (voxel march, 10 % occupancy, all channels); cfg0: 256 rays x 64 samples, hash grid, rgb."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

mode = sys.argv[1] if len(sys.argv) > 1 else "voxel"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda:0")
gflag = ["--graphs", os.environ.get("BENCH_GRAPHS", "on")]
if mode in ("voxel", "voxelrgb"):
    args = bench.parse(["--rays", "4096", "--raymarch", "voxel"] + gflag)
    nef, tracer = bench.make_model(args, dev, 0), bench.make_tracer(args, "voxel")
    bench.synthetic_prune(nef, 0.1)
    chans = {"rgb", "depth", "semantics", "inst_embedding"} if mode == "voxel" else {"rgb"}
    n = 4096
else:
    args = bench.parse(["--rays", "256", "--samples", "64", "--grid", "hash"] + gflag)
    nef, tracer = bench.make_model(args, dev, 0, grid="hash"), bench.make_tracer(args, "ray", 64)
    chans = {"rgb"}
    n = 256
rays, gt = bench.make_rays(n, dev, 1)
opt = bench.make_optimizer(nef)
for _ in range(10):
    bench.train_step(nef, tracer, opt, rays, gt, chans, 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    bench.train_step(nef, tracer, opt, rays, gt, chans, 1)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print("%s: %.3f ms per step (wall)" % (mode, dt * 1e3))
# GPU-side time of the same steps
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
from pagnerf_amd import ops
tracer.use_graphs = False
ops.profile_start()
for _ in range(20):
    bench.train_step(nef, tracer, opt, rays, gt, chans, 1)
prof = ops.profile_stop()
tot = sum(sum(v) for v in prof.values()) / 20
print("C-ABI kernels: %.3f ms per step over %d calls per step" % (tot, sum(len(v) for v in prof.values()) // 20))
for k, v in sorted(prof.items(), key=lambda kv: -sum(kv[1])):
    print("   %-32s %2d calls  %.4f ms per step" % (k, len(v) // 20, sum(v) / 20))
tracer.use_graphs = args.graphs == "on"
print("graphs:", getattr(tracer, "_graphs", None) and vars(tracer._graphs).get("captures"), getattr(tracer, "_graphs", None) and tracer._graphs.replays,
      getattr(tracer, "_graphs", None) and tracer._graphs.overflows)
if len(sys.argv) > 3 and sys.argv[3] == "nocprofile":
    sys.exit(0)
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    bench.train_step(nef, tracer, opt, rays, gt, chans, 1)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
