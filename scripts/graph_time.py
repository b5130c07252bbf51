import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from pagnerf_amd import ops
dev = torch.device("cuda:0")
args = bench.parse(["--rays", "4096", "--raymarch", "voxel", "--graphs", "on"])
nef, tracer = bench.make_model(args, dev, 0), bench.make_tracer(args, "voxel")
bench.synthetic_prune(nef, 0.1)
chans = {"rgb", "depth", "semantics", "inst_embedding"}
rays, gt = bench.make_rays(4096, dev, 1)
opt = bench.make_optimizer(nef)
for _ in range(6):
    bench.train_step(nef, tracer, opt, rays, gt, chans, 1)
torch.cuda.synchronize()
st = next(iter(tracer._graphs.states.values()))
print("counts", list(st.counts), "caps", list(st.buckets), "buf.cap", st.buf.cap)
g = next(iter(st.buckets.values()))
def ev(fn, n=20):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
print("fwd graph replay %.3f ms" % ev(g.fwd.replay))
print("bwd graph replay %.3f ms" % ev(g.bwd.replay))
def march():
    mb, _ = ops.march_into(st.buf, rays.origins, rays.dirs, rays.dist_min, rays.dist_max, 2, occupancy_bits=nef.grid.blas_bits, blas_level=7, max_travel=6.0,
                           occupancy_coarse_bits=nef.grid._coarse_bits(nef.grid.blas_bits))
    st.buf.pad_to(list(st.buckets)[0])
    if mb is not None:
        ops._poll_count(mb); ops._release_mailbox(mb)
print("march+pad %.3f ms" % ev(march))
def step():
    bench.train_step(nef, tracer, opt, rays, gt, chans, 1)
print("full step (events) %.3f ms" % ev(step))
t0 = time.perf_counter()
for _ in range(50): step()
torch.cuda.synchronize()
print("full step (wall) %.3f ms" % ((time.perf_counter() - t0) / 50 * 1e3))
