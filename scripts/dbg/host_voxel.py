"""cProfile of whole training steps in the post-prune voxel regime (host-bound there: ~35 launches in 1.8 ms)."""
import cProfile, os, pstats, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
args = bench.parse(["--raymarch", "voxel", "--occupancy", "0.1", "--no-aux", "--no-cpu-baseline"])
dev = torch.device("cuda:0")
nef = bench.make_model(args, dev, 0)
tracer = bench.make_tracer(args, "voxel", args.samples)
bench.synthetic_prune(nef, args.occupancy)
rays, gt = bench.make_rays(args.rays, dev, 1)
opt = bench.make_optimizer(nef)
chans = ["rgb", "semantics", "inst_embedding", "depth"]
for _ in range(10):
    bench.train_step(nef, tracer, opt, rays, gt, chans, 1)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    bench.train_step(nef, tracer, opt, rays, gt, chans, 1)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
