"""cProfile of the host side of one validation image (batch_render, voxel march on the pruned grid, render_batch 8000): where do the ~400 us of Python per pack go?
usage: python scripts/host_profile_render.py [rgbd]"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import pagnerf_amd

args = bench.parse([])
dev = torch.device("cuda:0")
nef = bench.make_model(args, dev, seed=0).eval()
bench.synthetic_prune(nef, args.occupancy)
tracer = pagnerf_amd.PanopticPackedRFTracer(raymarch_type="voxel", num_steps=2, bg_color="white", ray_max_travel=6.0, use_graphs=False)
pipe = pagnerf_amd.Pipeline(nef, tracer)
rays, _ = bench.make_rays(720 * 1280, dev, seed=77)
chans = ["depth", "rgb"] if "rgbd" in sys.argv else ["depth", "inst_embedding", "rgb", "semantics"]
with torch.no_grad():
    pagnerf_amd.batch_render(pipe, rays, channels=chans, render_batch=8000)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    pagnerf_amd.batch_render(pipe, rays, channels=chans, render_batch=8000)
    torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats(25)
