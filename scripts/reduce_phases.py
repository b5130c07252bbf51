"""Where reduce_kernel's time goes, from in-kernel stamps (experiment build only):
    bash scripts/build_variant.sh rtime encode "-DPAG_REDUCE_TIMING"
    PAG_LIB_VARIANT=rtime python3 scripts/reduce_phases.py
Runs the default bench workload's train step a few times (eager, so that the LAST reduce launch of a step is the delta grid's), reads the
per-block stamps of the last launch and prints, per level: block start (relative to the first block), duration, and the split over the
phases (the per-group phases are those of a wave's FIRST group of 64 tiles; `more_grps` = the groups after it); then the launch's
occupancy timeline and the blocks a few CUs ran."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from pagnerf_amd import _lib

dev = torch.device("cuda:0")
args = bench.parse(["--graphs", "off"] + sys.argv[1:])          # e.g. --raymarch voxel --channels rgb : the post-prune regime
nef, tracer = bench.make_model(args, dev, 0), bench.make_tracer(args, args.raymarch)
if args.raymarch == "voxel":
    bench.synthetic_prune(nef, args.occupancy)
rays, gt = bench.make_rays(args.rays, dev, 1)
opt = bench.make_optimizer(nef)
chans = {"rgb", "depth", "semantics", "inst_embedding"} if args.channels == "all" else {"rgb"}
for _ in range(5):
    bench.train_step(nef, tracer, opt, rays, gt, chans, 1)
torch.cuda.synchronize()
lib = _lib.load()
fn = lib.pag_debug_reduce_times
fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
buf = np.zeros((8192, 2, 16), dtype=np.uint64)
assert fn(buf.ctypes.data, buf.nbytes) == 0
print("runtime occupancy of reduce_kernel<2,4,packed>: %s" % {(th, dy): lib.pag_debug_reduce_occupancy(th, dy) for th in (1024, 512) for dy in (65536, 32768)})
L = nef.grid.tables.shape[0]
nb = int((buf[:, 0, 13] > 0).sum())
NS = nb // L
print("blocks %d = %d levels x %d slices" % (nb, L, NS))
raw = buf[:nb].astype(np.int64)
# shader-clock stamps -> microseconds on the shared 100 MHz axis (per wave: its own clock ratio)
span_clk = (raw[:, :, 6] - raw[:, :, 0]).astype(np.float64)
span_rt = (raw[:, :, 14] - raw[:, :, 13]).astype(np.float64)
done = (raw[:, :, 6] > 0) & (span_rt > 0)
print("shader clock: %.0f MHz (median)" % (100.0 * np.median(span_clk[done] / span_rt[done])))
us_per_tick = np.where(done, span_rt / np.maximum(span_clk, 1) / 100.0, 0.0)
t0 = raw[:, 0, 13][done[:, 0]].min()


def us(k):          # stamp k of every (block, wave) in us since the first block started
    return (raw[:, :, 13] - t0) / 100.0 + (raw[:, :, k] - raw[:, :, 0]) * us_per_tick


order = [0, 1, 2, 7, 3, 9, 4, 5, 6]
names = ["lvl_max", "zero", "header", "prologue", "loop", "more_grps", "wait", "write"]
T = np.stack([us(k) for k in order], -1)                      # [nb, 2, 9]
print("times in us, mean over the level's slices; per wave: w0 | w15")
print("%5s %7s %7s | %s | %7s" % ("level", "start", "dur", " ".join("%9s" % n for n in names), "entries"))
lvl_of = raw[:, 0, 11]
for lv in range(L):
    sel = lvl_of == lv
    for w in (0, 1):
        ok = sel & done[:, w] & (raw[:, w, 9] > 0)
        if not ok.any():
            if w == 0:
                print("%5d   (no gradient / no entries)" % lv)
            continue
        tt = T[ok, w]
        ph = np.diff(tt, axis=1).mean(0)
        print("%5d %7.1f %7.1f | %s | %7d  %s" % (lv, tt[:, 0].mean(), (tt[:, -1] - tt[:, 0]).mean(), " ".join("%9.2f" % p for p in ph),
                                                 raw[ok, w, 10].mean(), "w0" if w == 0 else "w15"))
dur = T[:, 0, -1] - T[:, 0, 0]
print("longest blocks (wave 0's view):")
for i in np.argsort(-np.where(done[:, 0], dur, 0))[:12]:
    print("   level %2d slice %2d: start %6.1f dur %6.1f | %s | first group's entries %d" % (raw[i, 0, 11], raw[i, 0, 12], T[i, 0, 0], dur[i],
          " ".join("%8.2f" % p for p in np.diff(T[i, 0])), raw[i, 0, 10]))
print("per level: mean / max block duration")
print("   " + "  ".join("L%d %.0f/%.0f" % (lv, dur[(lvl_of == lv) & done[:, 0]].mean(), dur[(lvl_of == lv) & done[:, 0]].max()) for lv in range(L) if ((lvl_of == lv) & done[:, 0]).any()))
okb = done[:, 0]
start, end = T[okb, 0, 0], T[okb, 0, -1]
print("launch: last block ends at %.1f us; blocks resident over time:" % end.max())
for x in np.arange(0, end.max(), end.max() / 16):
    print("  t=%6.1f  resident %4d" % (x, int(((start <= x) & (end > x)).sum())))
# placement: which CU every block ran on (HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]; XCC_ID [3:0])
hw = buf[:nb, 0, 15]
cu = ((hw >> np.uint64(32)) & np.uint64(0xF)).astype(np.int64) * 256 + ((hw >> np.uint64(8)) & np.uint64(0xFF)).astype(np.int64)
cus = np.unique(cu[okb])
print("distinct CUs used: %d" % len(cus))
for c in cus[:6]:
    idx = np.nonzero(okb & (cu == c))[0]
    idx = idx[np.argsort(T[idx, 0, 0])]
    print("  cu %5d: " % c + " ".join("[L%d %.0f-%.0f]" % (raw[i, 0, 11], T[i, 0, 0], T[i, 0, -1]) for i in idx))
