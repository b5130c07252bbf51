"""Where bin_kernel's time goes, from in-kernel stamps (experiment build only):
    bash scripts/build_variant.sh btime encode "-DPAG_BIN_TIMING"
    PAG_LIB_VARIANT=btime python3 scripts/bin_phases.py
Mean over the workgroups of the last bin launch of one eager train step of the default bench workload, for wave 0 and wave 15."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from pagnerf_amd import _lib

dev = torch.device("cuda:0")
args = bench.parse(["--graphs", "off"])
nef, tracer = bench.make_model(args, dev, 0), bench.make_tracer(args, "ray")
rays, gt = bench.make_rays(args.rays, dev, 1)
opt = bench.make_optimizer(nef)
for _ in range(5):
    bench.train_step(nef, tracer, opt, rays, gt, {"rgb", "depth", "semantics", "inst_embedding"}, 1)
torch.cuda.synchronize()
lib = _lib.load()
fn = lib.pag_debug_bin_times
fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
buf = np.zeros((32768, 2, 16), dtype=np.uint64)
assert fn(buf.ctypes.data, buf.nbytes) == 0
ok = buf[:, 0, 13] > 0
raw = buf[ok].astype(np.int64)
print("workgroups stamped:", raw.shape[0])
order = [0, 1, 2, 3, 4, 6, 7, 8, 9, 10, 11]
names = ["loads", "level0", "level1", "level2", "barrier", "prefix", "place0", "write0", "barrier0", "levels1,2+hdr"]
clk = (raw[:, :, 11] - raw[:, :, 0]).astype(np.float64)
rt = (raw[:, :, 14] - raw[:, :, 13]).astype(np.float64)
print("shader clock %.0f MHz; workgroup duration %.2f us (wave 0)" % (100 * np.median(clk[:, 0] / rt[:, 0]), rt[:, 0].mean() / 100))
for w in (0, 1):
    t = raw[:, w][:, order].astype(np.float64)
    d = np.diff(t, axis=1) * (rt[:, w] / np.maximum(clk[:, w], 1))[:, None] / 100.0
    print("wave %2d: " % (15 * w) + "  ".join("%s %.2f" % (n, v) for n, v in zip(names, d.mean(0))) + "   (us)")
