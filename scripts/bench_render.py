"""Forward-only (validation render, trainer.py:637-649) throughput of the bench scene: rays/s with torch.no_grad().
usage: python3 scripts/bench_render.py [rays] [samples]"""
import os
import sys
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pagnerf_amd import ops

rays_n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
samples = int(sys.argv[2]) if len(sys.argv) > 2 else 512
args = bench.parse(["--rays", str(rays_n), "--samples", str(samples), "--grid", "permuto", "--precision", "bf16"])
dev = torch.device("cuda:0")
nef, tracer = bench.make_model(args, dev, 0), bench.make_tracer(args)
rays, _ = bench.make_rays(rays_n, dev, 1)
cases = (["rgb", "semantics", "inst_embedding", "depth"], ["rgb", "depth"])
if os.environ.get("RENDER_ONLY_ALL"):
    cases = cases[:1]
for chans in cases:
    with torch.no_grad():
        for _ in range(5):
            tracer(nef, channels=chans, rays=rays, stage="val")
        torch.cuda.synchronize()
        if not os.environ.get("RENDER_NO_PROFILE"):
            ops.profile_start()
        t0 = time.perf_counter()
        n = 30
        for _ in range(n):
            tracer(nef, channels=chans, rays=rays, stage="val")
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        prof = ops.profile_stop() if not os.environ.get("RENDER_NO_PROFILE") else {}
    print("%s: %.3f ms per %d-ray chunk = %.0f k rays/s" % ("+".join(chans), dt * 1e3, rays_n, rays_n / dt / 1e3))
    print("   " + ", ".join("%s %.3f" % (k.replace("pag_", ""), sum(v) / n) for k, v in sorted(prof.items()) if sum(v) / n > 0.02))
