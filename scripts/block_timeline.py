"""Per-workgroup timeline of the step's big kernels (experiment build only):
    bash scripts/build_variant.sh bt encode,mlp "-DPAG_BLOCK_TIMING"
    PAG_LIB_VARIANT=bt python3 scripts/block_timeline.py
Every workgroup of an instrumented kernel records start / end on the chip-wide 100 MHz clock (csrc/blocktime.h).  For the LAST launch of each
kernel in one eager train step of the default bench workload this prints: workgroups, span (first start -> last end), mean / max workgroup
time, peak residency, fill = sum of workgroup times / (span x peak residency) - what a launch loses to its ramp and its tail -, when
residency falls under half of its peak, and the residency curve."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from pagnerf_amd import _lib

SLOTS, BLOCKS = 8, 32768
NAMES = {"encode": ["bin_kernel", "permuto_fwd_kernel", "permuto_fwd_add_kernel", "reduce_kernel"],
         "mlp": ["mlp_fwd_fast", "mlp_fwd_wide_stats", "mlp_bwd_fused", "mlp_bwd_pair", "mlp_bwd_wide_blocks", "head_composite_fwd", "mlp_wgrad"]}
dev = torch.device("cuda:0")
args = bench.parse(["--graphs", "off"] + sys.argv[1:])
nef, tracer = bench.make_model(args, dev, 0), bench.make_tracer(args, "ray")
rays, gt = bench.make_rays(args.rays, dev, 1)
opt = bench.make_optimizer(nef)
chans = {"rgb", "depth", "semantics", "inst_embedding"}
lib = _lib.load()


def read(tu):
    fn = getattr(lib, "pag_debug_block_times_" + tu)
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    t = np.zeros((SLOTS, BLOCKS, 3), dtype=np.uint64)
    g = np.zeros((SLOTS, 4), dtype=np.uint32)
    assert fn(t.ctypes.data, g.ctypes.data) == 0
    return t, g


for _ in range(4):
    bench.train_step(nef, tracer, opt, rays, gt, chans, 1)
torch.cuda.synchronize()
for tu in NAMES:
    read(tu)                      # clears the tables
bench.train_step(nef, tracer, opt, rays, gt, chans, 1)
torch.cuda.synchronize()
for tu, names in NAMES.items():
    t, g = read(tu)
    for slot, name in enumerate(names):
        ok = t[slot, :, 0] > 0
        n = int(ok.sum())
        if n == 0:
            continue
        st = t[slot, ok, 0].astype(np.int64)
        en = t[slot, ok, 1].astype(np.int64)
        t0 = st.min()
        st, en = (st - t0) / 100.0, (en - t0) / 100.0
        span = en.max()
        dur = en - st
        grid = np.linspace(0, span, 400, endpoint=False)
        res = ((st[None, :] <= grid[:, None]) & (en[None, :] > grid[:, None])).sum(1)
        peak = int(res.max())
        fill = dur.sum() / (span * peak)
        after_peak = np.nonzero(res >= 0.5 * peak)[0]
        half_t = grid[after_peak[-1]] if len(after_peak) else 0.0
        hw = t[slot, ok, 2]
        cu = ((hw >> np.uint64(32)) & np.uint64(0xF)).astype(np.int64) * 256 + ((hw >> np.uint64(8)) & np.uint64(0xFF)).astype(np.int64)
        print("%-24s grid %s x %d thr: %5d workgroups%s, span %7.1f us, workgroup mean %6.1f max %6.1f us, peak residency %4d (%.2f / CU), fill %.2f, under half residency after %6.1f us (%.0f %% of the span)"
              % (name, tuple(int(x) for x in g[slot, :3]), int(g[slot, 3]), n, " (table full)" if n == BLOCKS else "", span, dur.mean(), dur.max(), peak, peak / max(1, len(np.unique(cu))), fill,
                 half_t, 100 * (1 - half_t / span)))
        bars = " ".join("%4d" % int(r) for r in res[::25])
        print("     residency every %.1f us: %s" % (span / 16, bars))
        # per CU: how many workgroups it holds at once (time-weighted mean / max over the middle half), and how long a freed place stays empty
        lo, hi = 0.25 * span, 0.75 * span
        means, maxes, gaps = [], [], []
        for c in np.unique(cu):
            sel = cu == c
            ev = sorted([(a, 1) for a in st[sel]] + [(b, -1) for b in en[sel]])
            cur, last, area, mx, freed = 0, lo, 0.0, 0, None
            for tt, d in ev:
                if tt > hi:
                    break
                if tt >= lo:
                    area += cur * (tt - last)
                    last = tt
                    if d < 0:
                        freed = tt
                    elif freed is not None:
                        gaps.append(tt - freed)
                        freed = None
                cur += d
                if tt >= lo:
                    mx = max(mx, cur)
            means.append(area / max(last - lo, 1e-9))
            maxes.append(mx)
        print("     per CU over the middle half: mean workgroups resident %.2f (min %.2f max %.2f over CUs), most at once %d..%d; end of one workgroup -> next start on that CU: median %.2f us, mean %.2f us"
              % (np.mean(means), np.min(means), np.max(means), min(maxes), max(maxes), np.median(gaps) if gaps else 0, np.mean(gaps) if gaps else 0))
        xcc = ((hw >> np.uint64(32)) & np.uint64(0xF)).astype(np.int64)
        mid = (grid > 0.25 * span) & (grid < 0.75 * span)
        per = []
        for x in np.unique(xcc):
            sel = xcc == x
            r = ((st[None, sel] <= grid[mid, None]) & (en[None, sel] > grid[mid, None])).sum(1).mean()
            per.append("%d: %d wg, mean %.1f us, resident %.0f" % (x, int(sel.sum()), dur[sel].mean(), r))
        print("     per XCD (residency over the middle half of the span): " + " | ".join(per))
