#!/usr/bin/env python3
"""rays/s of one PAg-NeRF train step through the HIP hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--channels all|rgb] [--rays 4096] [--samples 512]
                    [--grid permuto|hash] [--raymarch ray|voxel --occupancy 0.1] [--pose-opt] [--dry-run]

Workload at N = 1 (BASELINE.json configs[1]): PanopticDeltaNeF + permutohedral grids (L = 24, F = 2,
T = 2^18, main + delta), 4096 rays x 512 'ray'-mode samples (M = 2 097 152 packed samples, dense
occupancy), bf16 MFMA decoders on bf16 features with fp32 tables / accumulation / compositing.
A step = ray march -> encode -> decoders -> compositing -> loss (rgb L1 x10, + semantic / instance
NLL against fixed synthetic labels when the panoptic heads are on; pc_nerf/trainer.py:443-480) ->
backward through every kernel -> Adam (eps 1e-15, grid lr x100) - nothing is cached between steps.

N > 1: one rank per GPU over RCCL.  Launched either by the driver (`python -m torch.distributed.run
--nproc-per-node N bench.py --gpus N ...`) or by this script itself: with `--gpus N` and no
WORLD_SIZE in the environment the parent process - before anything touches a GPU - starts the N
ranks as a child `torch.distributed.run`, relays rank 0's JSON line and exits with the child's code.
Every rank marches its own 4096-ray shard against replicated parameters (weak scaling); gradients are
averaged with shard.GradSync (delta table early + one flat all-reduce); value = all ranks' rays /
max-over-ranks time.  `rccl_ranks_seen` is an all-reduce of ones over the process group and must
equal N, as must `n_gpus`, or the run fails.  Next to the weak line the N > 1 run reports
`strong` (BASELINE configs[3]: 6 images x 4096 rays with ba_pipeline pose optimisation, the 24 576
rays split over the ranks) and `render_sharded` (validation render + the all_gather of the buffers).

Output: ONE compact JSON line (<= 4 KB, compact_line(): the contract's keys, `roofline` and `cpu_baseline` as flat objects, one number
per extra measurement) as the LAST stdout line; the full record - every block below with its notes and per-entry-point tables - goes
to bench_detail.json beside this file (and to gpurun_out/ when that exists; PAG_BENCH_DETAIL overrides the path), a digest to stderr.

The record carries
  roofline      the permutohedral encode forward launch (the grid-interpolate kernel north_star sets
                the 40 % HBM target on): algorithmic bytes per launch / its mean duration measured
                with HIP events on the launch stream inside the timed region
  kernels       per C-ABI entry point: ms per step, algorithmic bytes, fraction of the HBM roof;
                mfma: algorithmic decoder FLOP/s against the dense bf16 MFMA peak
  sustained     the same step for >= 2 s (--sustain-steps, default 300): ms/step overall and over the last half
  configs       (N = 1) the other single-GPU BASELINE configurations, each a short run
  cpu_baseline  the CPU oracle's restatement of the reference's grids/hash_grid_torch.py path
                (encode -> decoders -> compositing, forward + backward) on this host's cores,
                on a bounded sample of the same workload (kind "port").
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_PEAK_TFLOPS = 2500.0    # same guide: dense bf16 MFMA ~2.5 PFLOP/s
SETTLE_STEPS = 6             # untimed steps between the interpreter's garbage collection and the barrier that opens a timed region (Job.timed)
LINE_BUDGET = 4096           # the driver parses the LAST stdout line: round 5's grew to 32 KB and was recorded as `parsed: null`


# --------------------------------------------------------------------------------------------- the result line and its detail file
def _short(text, n):
    text = str(text)
    return text if len(text) <= n else text[:n - 3] + "..."


def _config_key(name):
    """'configs[4] per-GPU shard: 131072 rays x 64 samples, permuto, fp16 tables + ...' -> a short unique key for the one-number-per-config map."""
    head, _, rest = str(name).partition(":")
    head = head.strip()
    words = [w for w in rest.replace(",", " ").split() if w in ("hash", "permuto", "fp16", "fp32", "rgb", "all")]
    return _short(head + ("/" + "+".join(dict.fromkeys(words)) if words else ""), 56)


def compact_line(detail, detail_path=None):
    """The ONE line the driver parses: the contract's scalars, `config`, `roofline` and `cpu_baseline` as flat objects, and one number per
    extra measurement.  Everything else (per-entry-point tables, notes, per-regime breakdowns) lives in the detail file written beside
    it.  Always <= LINE_BUDGET bytes: optional blocks are dropped, last first, if a run ever grows past it."""
    out = {k: detail.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                      "vs_baseline", "dtype", "data")}
    cfg = dict(detail.get("config") or {})
    out["config"] = {k: (_short(v, 360) if isinstance(v, str) else v) for k, v in cfg.items() if isinstance(v, (str, int, float, bool, list)) or v is None}
    out["rccl_ranks_seen"], out["backend"] = detail.get("rccl_ranks_seen"), detail.get("backend")
    gs = detail.get("grad_sync")
    out["grad_sync"] = (gs.get("comm_dtype") if isinstance(gs, dict) else gs)
    if detail.get("dry_run"):
        out["dry_run"] = True
    rf = detail.get("roofline")
    if isinstance(rf, dict):
        keep = {k: v for k, v in rf.items() if isinstance(v, (int, float, bool)) or v is None or k in ("bound", "unit", "kernel")}
        ts = rf.get("traffic_source")
        keep["traffic_source"] = (ts.get("file") if isinstance(ts, dict) else ts)
        out["roofline"] = keep
    else:
        out["roofline"] = None
    cb = detail.get("cpu_baseline")
    if isinstance(cb, dict):
        out["cpu_baseline"] = {k: (_short(cb.get(k), 300) if k == "sample" else cb.get(k)) for k in ("value", "unit", "cores", "cores_available", "kind", "sample")}
    else:
        out["cpu_baseline"] = None
    optional = []          # (key, value), most important first

    def num(block, *keys):
        b = detail.get(block)
        if isinstance(b, dict):
            got = {k: b[k] for k in keys if isinstance(b.get(k), (int, float))}
            if got:
                optional.append((block, got))
    num("mfma_util", "frac")
    num("schedule_weighted", "ms_per_step", "rays_s")
    num("sustained", "ms_per_step")
    if isinstance(detail.get("configs"), list):
        optional.append(("configs", {(c.get("key") or _config_key(c.get("name"))): c.get("ms_per_step") for c in detail["configs"] if isinstance(c, dict)}))
    by = (detail.get("best_yaml_step") or {}).get("regimes") if isinstance(detail.get("best_yaml_step"), dict) else None
    if isinstance(by, dict):
        optional.append(("best_yaml_step", {k: v.get("ms_per_step") for k, v in by.items() if isinstance(v, dict)}))
    num("rgb_only", "ms_per_step")
    num("eager", "ms_per_step")
    la = detail.get("with_lin_assignment")
    if isinstance(la, dict):
        optional.append(("with_lin_assignment", {k: v.get("ms_per_step") for k, v in la.items() if isinstance(v, dict)}))
    for blk in ("render", "render_pruned"):
        b = detail.get(blk)
        if isinstance(b, dict):
            optional.append((blk, {k: v.get("ms_per_image") for k, v in b.items() if isinstance(v, dict) and "ms_per_image" in v}))
    if isinstance(detail.get("weak_regimes"), list):
        def _weak(w):
            e = dict(ms_per_step=w.get("ms_per_step"), grad_sync=((w.get("grad_sync") or {}).get("comm_dtype") if isinstance(w.get("grad_sync"), dict) else w.get("grad_sync")))
            if "exchanged_bytes" in w:
                e.update(exchanged_bytes=w["exchanged_bytes"], dense_bytes=w.get("dense_bytes"))
            elif isinstance(w.get("sparse_sync"), list):
                e.update(exchanged_bytes=sum(t["exchanged_bytes"] for t in w["sparse_sync"]), dense_bytes=sum(t["dense_bytes"] for t in w["sparse_sync"]))
            return e
        optional.append(("weak_regimes", {_short(w.get("name"), 90): _weak(w) for w in detail["weak_regimes"] if isinstance(w, dict)}))
    num("strong", "ms_per_step", "rays_s")
    num("render_sharded", "ms", "rays_s")
    kt = detail.get("kernels")
    if isinstance(kt, dict):
        optional.append(("kernels_ms", {k: v.get("ms_per_step") for k, v in kt.items() if isinstance(v, dict) and (v.get("ms_per_step") or 0) >= 0.02}))
    if detail_path:
        out["detail"] = detail_path
    for k, v in optional:
        out[k] = v
    for k, _ in reversed(optional):
        if len(json.dumps(out)) <= LINE_BUDGET:
            break
        del out[k]
    assert len(json.dumps(out)) <= LINE_BUDGET, "bench.py: the result line cannot be brought under %d bytes" % LINE_BUDGET
    return out


def emit(detail):
    """Write the full record to bench_detail.json (repo root; also gpurun_out/ when that directory exists; PAG_BENCH_DETAIL overrides the
    path), a per-key digest of it to stderr, and the compact line - LAST, alone - to stdout."""
    path = os.environ.get("PAG_BENCH_DETAIL") or os.path.join(ROOT, "bench_detail.json")
    paths = [path]
    scratch = os.path.join(ROOT, "gpurun_out")
    if "PAG_BENCH_DETAIL" not in os.environ and os.path.isdir(scratch):
        paths.append(os.path.join(scratch, "bench_detail.json"))
    written = None
    for p_ in paths:
        try:
            with open(p_, "w") as f:
                json.dump(detail, f, indent=1)
            written = written or p_
        except OSError as e:
            print("bench.py: could not write %s: %s" % (p_, e), file=sys.stderr)
    for k, v in detail.items():
        if isinstance(v, (dict, list)):
            print("[bench detail] %s: %s" % (k, _short(json.dumps(v), 700)), file=sys.stderr)
    sys.stderr.flush()
    line = compact_line(detail, os.path.relpath(written, ROOT) if written and written.startswith(ROOT) else written)
    print(json.dumps(line), flush=True)
    return line


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--channels", default="all", choices=["all", "rgb", "rgbd"], help="rgbd = rgb + depth (what best.yaml's inst_outlier_rejection adds to every step)")
    ap.add_argument("--lin-assign", action="store_true",
                    help="with --pose-opt --channels all: the instance term as best.yaml forms it (LinAssignmentThingsLoss(outlier_rejection=True) over "
                         "--images images + segment_consistency_regularizer) instead of fixed instance targets")
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--samples", type=int, default=512)
    ap.add_argument("--grid", default="permuto", choices=["permuto", "hash"])
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--raymarch", default="ray", choices=["ray", "voxel"])
    ap.add_argument("--occupancy", type=float, default=0.1, help="voxel mode: fraction of occupied 128^3 cells after the synthetic prune")
    ap.add_argument("--pose-opt", action="store_true", help="BAPipeline: rays from learnable camera extrinsics (configs[3])")
    ap.add_argument("--two-call", action="store_true", help="with --lin-assign: the begin / rgb backward / finish / rest backward form (INTEGRATION.md)")
    ap.add_argument("--images", type=int, default=6, help="--pose-opt: images per step (rays are split evenly over them)")
    ap.add_argument("--table-dtype", default="fp32", choices=["fp32", "fp16"], help="grid tables (BASELINE configs[4]: fp16 features)")
    ap.add_argument("--fp32-coords", action="store_true", help="permuto grids: skip the fp16 coordinate rounding of the reference's autocast")
    ap.add_argument("--sustain-steps", type=int, default=300, help="extra timed region after the K steps (0 = off)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-aux", action="store_true", help="skip the rgb-only / sustained / configs / strong measurements")
    ap.add_argument("--graphs", default="on", choices=["on", "off", "static"],
                    help="PanopticPackedRFTracer(use_graphs=...): on = replay the post-march part of the step as HIP graphs (no pose-opt; at N > 1 the "
                         "backward is captured as two graphs so the delta table's all-reduce starts between them); static = static padded buffers + "
                         "optimistic count check with eager launches; off = eager")
    ap.add_argument("--grad-sync", default="auto", choices=["auto", "fp32", "bf16"],
                    help="N > 1: table gradients as RCCL fp32 all-reduce, as bf16 messages with fp32 accumulation (shard._DirectReduce), or (default) chosen "
                         "by regime: bf16 when the measured step is shorter than 4 x the predicted exposed fp32 exchange (shard.GradSync(comm_dtype='auto'))")
    ap.add_argument("--sparse-sync", default="off", choices=["off", "bounded", "exact"],
                    help="N > 1: table gradients travel as the union of the ranks' touched rows (shard.SparseRows): bounded = slots sized from earlier steps, the "
                         "host never waits; exact = the host reads the per-level counts each step.  The --dry-run regime lines always run it (bounded unless said otherwise)")
    ap.add_argument("--dry-run", action="store_true", help="CPU + gloo: process group, shard collectives, timing and JSON plumbing only")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------ self-launch (parent, never touches a GPU)
def spawn_ranks(args, argv):
    """`--gpus N` without WORLD_SIZE: run N fresh ranks under torch.distributed.run as a CHILD process and relay rank 0's line."""
    if not args.dry_run:
        from pagnerf_amd import build as b       # hipcc only (no HIP runtime call, no dlopen of the library): every rank finds it built
        b.build(verbose=True)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for out in proc.stdout:
        out = out.rstrip("\n")
        if out.startswith('{"metric"'):
            line = out
        else:
            print(out, file=sys.stderr)
    rc = proc.wait()
    if rc != 0:
        print("bench.py: the %d-rank child run failed with exit code %d" % (args.gpus, rc), file=sys.stderr)
        return rc or 1
    if line is None:
        print("bench.py: the child run printed no result line", file=sys.stderr)
        return 1
    got = json.loads(line)
    if got.get("n_gpus") != args.gpus or got.get("rccl_ranks_seen") != args.gpus:
        print("bench.py: asked for %d ranks, the run reports n_gpus=%s rccl_ranks_seen=%s" %
              (args.gpus, got.get("n_gpus"), got.get("rccl_ranks_seen")), file=sys.stderr)
        return 1
    print(line)
    return 0


# ------------------------------------------------------------------------------------------------ workload
def make_model(args, dev, seed, grid=None, num_lods=None, log2T=None, finest=None, table_dtype=None, heads=None):
    import torch
    import pagnerf_amd
    grid = grid or args.grid
    torch.manual_seed(seed)
    tdt = torch.float16 if (table_dtype or args.table_dtype) == "fp16" else torch.float32
    sem_l, inst_l = heads or (1, 2)        # best.yaml:92,74; (2, 1) = lin_assign_delta_app.yaml / lin_assign_direct_app.yaml / contrastive_delta_app.yaml
    common = dict(feature_dim=2, num_classes=6, num_instances=200, sem_num_layers=sem_l, sem_softmax=True, inst_num_layers=inst_l,
                  inst_softmax=True, panoptic_features_type="delta", hidden_dim=64, num_layers=1, view_multires=4,
                  precision=args.precision, blas_level=7, table_dtype=tdt)
    if grid == "permuto":        # configs/bup20/best.yaml:47-65
        cap = log2T or 18
        nef = pagnerf_amd.PanopticDeltaNeF(grid_type="PermutoGrid", num_lods=24, capacity_log_2=cap, delta_capacity_log_2=cap,
                                           coarsest_scale=1.0, finest_scale=1e-4, half_coords=not args.fp32_coords, **common)
        for g in (nef.grid, nef.delta_grid):
            g.init_from_scales(tables=torch.randn(24, 2 ** cap, 2) * 1e-2)
    else:                        # BASELINE.json configs[2]: 16-level hash grid, T = 2^19
        L_ = num_lods or 16
        nef = pagnerf_amd.PanopticDeltaNeF(grid_type="HashGridTorch", num_lods=L_, codebook_bitwidth=log2T or 19, **common)
        for g in (nef.grid, nef.delta_grid):
            g.init_from_resolutions([16] * (L_ - 1) + [finest or 2048])
            g.tables.data.copy_((torch.randn(g.tables.shape) * 1e-2).to(g.tables.dtype))
    return nef.to(dev)


def make_tracer(args, raymarch=None, samples=None):
    import pagnerf_amd
    rm = raymarch or args.raymarch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # at N > 1 the tracer captures the backward as TWO graphs (panoptic heads | the rest, pagnerf_amd/graphs.py) behind two autograd nodes, so
    # shard.GradSync's post-accumulate hook on the delta table - its early all-reduce - fires between them as in an eager backward;
    # --graphs static keeps the uncaptured static-buffer form (ordinary autograd pass, host never waits for the sample count)
    g = {"on": True, "static": "static", "off": False}[args.graphs]
    if rm == "voxel":            # after trainer.py:362-366: 2 samples per intersected occupied voxel (best.yaml:31), ray_max_travel 6 x scale
        return pagnerf_amd.PanopticPackedRFTracer(raymarch_type="voxel", num_steps=2, bg_color="white", ray_max_travel=6.0, use_graphs=g)
    return pagnerf_amd.PanopticPackedRFTracer(raymarch_type="ray", num_steps=samples or args.samples, bg_color="white", use_graphs=g)


def synthetic_prune(nef, fraction, seed=0):
    """Occupancy bitfield of a 'plant row' blob covering `fraction` of the 128^3 cells (stands for nef.prune() at epoch 201)."""
    import torch
    g = nef.grid
    R = 2 ** g.blas_level
    ar = (torch.arange(R, dtype=torch.float32) + 0.5) / R * 2 - 1
    x, y, z = torch.meshgrid(ar, ar, ar, indexing="ij")
    gen = torch.Generator().manual_seed(seed)
    f = torch.zeros(R, R, R)
    for _ in range(24):          # sum of anisotropic bumps along the x axis ("row"), low in z
        c = torch.rand(3, generator=gen) * torch.tensor([1.6, 0.8, 0.8]) - torch.tensor([0.8, 0.4, 0.9])
        s = 0.08 + 0.2 * torch.rand(3, generator=gen)
        f += torch.exp(-(((x - c[0]) / s[0]) ** 2 + ((y - c[1]) / s[1]) ** 2 + ((z - c[2]) / s[2]) ** 2))
    thr = torch.quantile(f.reshape(-1)[::7], 1.0 - fraction)
    mask = (f > thr).reshape(-1)
    for grid in (nef.grid, nef.delta_grid):
        grid.blas_init(mask.to(grid.blas_bits.device))
    return float(mask.float().mean())


def make_rays(n, dev, seed):
    """BUP20-shaped synthetic view: downward-looking pinhole rays that stay inside [-1,1]^3
    (near 0, datasets/formats/bup20.py:249; far scaled so every sample survives the dense BLAS)."""
    import torch
    import pagnerf_amd
    g = torch.Generator().manual_seed(seed)
    o = torch.cat([(torch.rand(n, 2, generator=g) - 0.5) * 0.6, torch.full((n, 1), 0.95)], 1)
    d = torch.cat([(torch.rand(n, 2, generator=g) - 0.5) * 0.7, -torch.ones(n, 1)], 1)
    d = torch.nn.functional.normalize(d, dim=-1)
    gt = dict(rgb=torch.rand(n, 3, generator=g), sem=torch.randint(0, 6, (n,), generator=g),
              inst=torch.randint(0, 200, (n,), generator=g))
    # per-ray gt instance ids for the linear-assignment loss: ~24 'plants' per image (id > 0) tiled over the rays' footprint, the
    # rest stuff (id 0); stuff mask = semantic class is a stuff class (trainer.py:493-494)
    cell = ((o[:, 0] + 0.3) / 0.6 * 6).long().clamp(0, 5) * 6 + ((o[:, 1] + 0.3) / 0.6 * 6).long().clamp(0, 5)
    gt["inst_ids"] = torch.where(cell % 3 != 0, cell + 1000, torch.zeros_like(cell))
    gt["stuff"] = gt["sem"] < 2
    return pagnerf_amd.Rays(o.to(dev), d.to(dev), dist_min=0.0, dist_max=1.9), {k: v.to(dev) for k, v in gt.items()}


def sphere_gt(origins, dirs):
    """Closed-form colours of a textured sphere of radius 0.5 at the origin under a white background for world-frame rays (the analytic scene of
    scripts/train_synthetic.py): a LEARNABLE scene - after a few hundred steps the model has empty space (sigma = relu(pre) = 0 exactly) around a
    surface, which is what real training batches look like and what the backward's zero-gradient early-outs act on."""
    import torch
    o, d = origins.detach().float().cpu(), dirs.detach().float().cpu()
    b = (o * d).sum(-1)
    disc = b * b - ((o * o).sum(-1) - 0.25)
    hit = disc > 0
    t = -b - torch.sqrt(disc.clamp_min(0))
    p = o + d * t[:, None]
    return torch.where(hit[:, None], 0.5 + 0.5 * torch.sin(p * 9.0 + torch.tensor([0.0, 2.0, 4.0])), torch.ones(o.shape[0], 3)), hit


class PoseOpt:
    """configs[3]: rays of `images` cameras generated from learnable extrinsics (pc_nerf/ba_pipeline.py:85-92).  The camera-frame base
    rays of the whole step are fixed; a rank transforms its own contiguous block with per-ray camera indices."""

    def __init__(self, nef, tracer, total_rays, images, dev, lo, hi, seed=7):
        import torch
        from pagnerf_amd.ba_pipeline import BAPipeline
        gen = torch.Generator().manual_seed(seed)
        C = images
        views = torch.eye(4).repeat(C, 1, 1)
        ang = (torch.rand(C, generator=gen) - 0.5) * 0.3                   # small yaw around z, camera above the scene looking down
        views[:, 0, 0], views[:, 0, 1], views[:, 1, 0], views[:, 1, 1] = torch.cos(ang), -torch.sin(ang), torch.sin(ang), torch.cos(ang)
        t = torch.cat([(torch.rand(C, 2, generator=gen) - 0.5) * 0.2, torch.full((C, 1), -0.95)], 1)    # o_w = R^T (o_c - t)
        views[:, :3, 3] = t
        self.pipe = BAPipeline(nef, views, tracer=tracer, anchor_frame_idxs=[0], near=0.0, far=1.9).to(dev)
        per = total_rays // C
        o = torch.zeros(total_rays, 3)
        d = torch.cat([(torch.rand(total_rays, 2, generator=gen) - 0.5) * 0.7, -torch.ones(total_rays, 1)], 1)
        cam = torch.arange(total_rays) // max(per, 1)
        self.o, self.d, self.cam = o[lo:hi].to(dev), d[lo:hi].to(dev), cam[lo:hi].clamp(max=C - 1).int().to(dev)

    def rays(self):
        return self.pipe.transform_rays_indexed(self.o, self.d, self.cam)

    def points_3d(self, depth):
        """pc_nerf/trainer.py:508-518 `rays_to_3d_points(rays, rb.depth, cameras)`: the step's base rays unprojected by the rendered depth."""
        return self.pipe.rays_to_3d_points_indexed(self.o, self.d, depth, self.cam)

    def parameters(self):
        return [self.pipe.camera_extrinsics]


def make_optimizer(nef, extra=()):
    import torch
    grid_params = [p for n, p in nef.named_parameters() if "grid" in n]
    rest = [p for n, p in nef.named_parameters() if "grid" not in n]
    groups = [dict(params=grid_params, lr=1e-3 * 100), dict(params=rest, lr=1e-3)]     # best.yaml:103,108 ; trainer.py:272-281
    if any(p.dtype == torch.float16 for p in grid_params):
        groups[0]["eps"] = 1e-4       # fp16 tables keep fp16 Adam state: the reference's eps = 1e-15 underflows to 0 there (0 / 0 on untouched rows)
    if extra:
        groups.append(dict(params=list(extra), lr=1e-4))
    if os.environ.get("PAG_BENCH_TORCH_ADAM") or any(p.dtype != torch.float32 for p in grid_params):
        # A/B switch, or fp16 tables (BASELINE configs[4]): pag_adam_step covers fp32 tensors - torch's own fused multi-tensor kernel
        try:
            return torch.optim.Adam(groups, eps=1e-15, fused=True)
        except Exception:
            return torch.optim.Adam(groups, eps=1e-15)
    # config_parser.py:667-673 `optim_cls(params, eps=1e-15)` with optim_cls = torch.optim.Adam: pagnerf_amd.optim.Adam is that class with its
    # step() on pag_adam_step (same arithmetic, same state); groups it does not cover (fp16 tables) take torch's implementation inside it
    import pagnerf_amd
    return pagnerf_amd.optim.Adam(groups, eps=1e-15)


class ReferenceFormulationThingsLoss:
    """loss/lin_assignment_things.py:23-82 as the reference writes it, on device tensors: one masked sum and one device-to-host
    copy PER gt label for the cost matrix (:31-33), one masked assignment per label for the relabelling (:47-50).  Here only to time
    that formulation beside pagnerf_amd.loss.LinAssignmentThingsLoss (one pag_label_sums launch, one [K,199] copy, one table lookup);
    both produce the same virtual labels (tests/test_gpu_loss.py pins the device form to the reference's golden labels)."""

    def __call__(self, inst_probabilities, labels_gt, stuff_mask):
        import numpy as np
        import scipy.optimize
        import torch
        import torch.nn.functional as F
        loss = torch.zeros_like(inst_probabilities[..., 0])
        for i, (p, gt, m) in enumerate(zip(inst_probabilities, labels_gt, stuff_mask)):
            valid = torch.logical_or(m, gt > 0)
            gt_v, p_v = gt[valid], p[valid]
            with torch.no_grad():
                things = gt_v > 0
                tg, tp = gt_v[things], p_v[things][..., 1:]
                labels = sorted(torch.unique(tg).cpu().tolist())[:tp.shape[-1]]
                cost = np.zeros([len(labels), tp.shape[-1]])
                for k, lab in enumerate(labels):
                    cost[k, :] = -(tp[tg == lab, :].sum(dim=0) / ((tg == lab).sum() + 1e-4)).cpu().numpy()
                rows, cols = scipy.optimize.linear_sum_assignment(np.nan_to_num(cost))
                tl = torch.zeros_like(tg)
                for a, r in enumerate(rows):
                    tl[tg == labels[r]] = int(cols[a])
                virt = torch.zeros_like(gt_v)
                virt[things] = tl + 1
            if torch.any(virt != p_v.argmax(dim=-1)):
                loss[i][valid] = F.nll_loss(torch.log(p_v + 1e-27), virt, reduction="none")
        return loss


def train_step(nef, tracer, opt, rays, gt, channels, world, sync=None, lin_assign=None, images=1, points_fn=None, seg_reg=False, overlap=False):
    """images / points_fn / seg_reg: the instance term over a batch of `images` images as pc_nerf/trainer.py:483-533 forms it under
    configs/bup20/best.yaml - LinAssignmentThingsLoss per image (outlier rejection from the rendered depth's 3-D points when points_fn is
    given, :508-518) + segment_consistency_regularizer on the same probabilities (:525-527, weight 1.0)."""
    import torch
    import torch.nn.functional as F
    opt.zero_grad(set_to_none=True)
    if callable(rays):
        rays = rays()            # pose optimisation: this step's rays from the current extrinsics
    rb = tracer(nef, channels=channels, rays=rays, stage="train")
    # trainer.py:443-446 / best.yaml:116 rgb L1; trainer.py:465-467 nll_loss(log(p + 1e-27), gt); the instance term stands for the
    # per-image linear-assignment NLL (trainer.py:499-520 -> loss/lin_assignment_things.py:80), same arithmetic on a fixed target.
    # pagnerf_amd.loss.render_loss evaluates exactly that sum in one launch (and its gradients in one more).
    from pagnerf_amd.loss import render_loss, NllTerm, segment_consistency_regularizer
    if os.environ.get("PAG_BENCH_TORCH_LOSS"):
        loss = 10.0 * torch.abs(rb.rgb - gt["rgb"]).mean()
        if "semantics" in channels:
            loss = loss + 0.1 * F.nll_loss(torch.log(rb.semantics + 1e-27), gt["sem"], reduction="mean")
            loss = loss + 1000.0 * F.nll_loss(torch.log(rb.inst_embedding + 1e-27), gt["inst"], reduction="mean")
    elif "semantics" in channels and lin_assign is not None:
        # the instance term as the trainer forms it late in training (trainer.py:483-533, best.yaml inst_loss linear_assignment_things):
        # per-image Hungarian relabelling of the rendered instance probabilities, then the NLL against the virtual labels
        B = images
        inst = rb.inst_embedding.float().reshape(B, -1, rb.inst_embedding.shape[-1])
        ids, stuff = gt["inst_ids"].reshape(B, -1), gt["stuff"].reshape(B, -1)
        pts = points_fn(rb.depth.detach()).reshape(B, -1, 3) if points_fn is not None else None
        if overlap and hasattr(lin_assign, "begin"):
            # two-call form: the assignment's launches and copies first, then everything that does not need its result - the
            # regulariser, the semantic term, and the colour / density / main-grid half of the backward (`loss_rgb.backward()`: the tracer's
            # graphs are split, graph_split=True) - is queued while the host waits for the cost matrices and runs SciPy
            pending = lin_assign.begin(inst, ids, stuff, pts)
            loss, _ = render_loss(rb.rgb, gt["rgb"], 10.0)
            loss.backward()
            reg = segment_consistency_regularizer(inst, ids, eps=1e-27) if seg_reg else None
            rest, _ = render_loss(term_a=NllTerm(rb.semantics, gt["sem"], weight=0.1))
            il = lin_assign.finish(pending)
            il = il.mean() if reg is None else il.mean() + 1.0 * reg
            rest = rest + 1000.0 * il
            rest.backward()
            loss = loss.detach() + rest.detach()
            if world > 1:
                sync.finish()
            opt.step()
            return loss
        loss, _ = render_loss(rb.rgb, gt["rgb"], 10.0, NllTerm(rb.semantics, gt["sem"], weight=0.1))
        reg = segment_consistency_regularizer(inst, ids, eps=1e-27) if seg_reg else None     # queued BEFORE the assignment's one synchronisation
        il = lin_assign(inst, ids, stuff, pts) if pts is not None else lin_assign(inst, ids, stuff)
        il = il.mean() if reg is None else il.mean() + 1.0 * reg                           # `inst_loss += w * reg` broadcasts the scalar over [B, P]; then .mean()
        loss = loss + 1000.0 * il
    elif "semantics" in channels:
        loss, _ = render_loss(rb.rgb, gt["rgb"], 10.0, NllTerm(rb.semantics, gt["sem"], weight=0.1),
                              NllTerm(rb.inst_embedding, gt["inst"], weight=1000.0))
    else:
        loss, _ = render_loss(rb.rgb, gt["rgb"], 10.0)
    loss.backward()
    if world > 1:
        sync.finish()       # delta-table all-reduce was launched from the backward; the rest goes as one flat RCCL all-reduce
    opt.step()
    return loss


def cpu_baseline(n_rays, n_samples, budget_s=9.0, points=((256, 64), (4096, 64)), point_budget_s=14.0):
    """The oracle's torch-CPU restatement of the reference's hash_grid_torch path on this host's cores (kind "port").
    `value`: full train step (forward + backward + Adam, rgb L1 x 10) on a bounded sample of the headline workload's shape
    (n_rays x n_samples).  `points`: the configurations BASELINE.md section 3 names - (256 rays x 64 samples) = BASELINE.json
    configs[0], (4096 x 64) - each in its three modes: encode forward, encode + decoders + compositing forward, full train step."""
    import numpy as np
    import torch
    from oracle import hash_encode as oh, decoders as od, render as orr
    # torch-CPU ops on these small tensors scale badly past a few dozen threads (256 threads on the GPU
    # box's host ran 400x slower than 8): use at most 32 and report both the count used and the count available.
    cores_available = os.cpu_count() or 1
    torch.set_num_threads(min(32, cores_available))
    rs = np.random.RandomState(0)
    L_, log2T = 16, 19
    res = oh.level_resolutions(16, 2048, L_)
    tables = torch.from_numpy(rs.uniform(-1e-4, 1e-4, size=(L_, 2 ** log2T, 2)).astype(np.float32)).requires_grad_(True)
    dims = {"density": (32, 64, 16), "color": (43, 64, 64, 3)}
    params = {}
    for k, d in dims.items():
        W = [torch.from_numpy((rs.standard_normal(size=(d[i + 1], d[i])) / np.sqrt(d[i])).astype(np.float32)).requires_grad_(True) for i in range(len(d) - 1)]
        b = [torch.zeros(d[i + 1], requires_grad=True) for i in range(len(d) - 1)]
        params[k] = (W, b)
    leaves = [tables] + [t for W, b in params.values() for t in W + b]
    opt = torch.optim.Adam([dict(params=[tables], lr=1e-1), dict(params=leaves[1:], lr=1e-3)], eps=1e-15)      # trainer.py:272-281, best.yaml:103,108

    def scene(nr):
        o = torch.cat([(torch.rand(nr, 2) - 0.5) * 0.6, torch.full((nr, 1), 0.95)], 1)
        dr = torch.nn.functional.normalize(torch.cat([(torch.rand(nr, 2) - 0.5) * 0.7, -torch.ones(nr, 1)], 1), dim=-1)
        return o, dr, torch.rand(nr, 3)

    def step(sc, ns, mode):
        o, dr, gt = sc
        nr = o.shape[0]
        with torch.set_grad_enabled(mode == "train"):
            ridx, pidx, samples, depths, deltas, boundary = orr.raymarch_ray(o, dr, 0.0, 1.9, ns, torch.rand(nr, ns))
            feats, _ = oh.hash_encode(samples[:, 0], tables, res, log2T)
            if mode == "encode":
                return
            out = od.nef_forward(feats, None, dr[ridx], params, {"rgb"})
            comp = orr.composite(nr, ridx, boundary, out["density"], deltas, rgb=out["rgb"])
            if mode == "forward":
                return
            opt.zero_grad(set_to_none=True)
            loss = 10.0 * torch.abs(comp["rgb"] - gt).mean()
            loss.backward()
            opt.step()

    def timed(sc, ns, mode, budget, max_steps=50):
        t_w = time.perf_counter()
        step(sc, ns, mode)                       # warm-up (also sizes the budget)
        t_w = time.perf_counter() - t_w
        if t_w > budget:                         # pathological host: report the single step rather than overrun
            return t_w, 1, True
        t0, n = time.perf_counter(), 0
        while True:
            step(sc, ns, mode)
            n += 1
            if time.perf_counter() - t0 > budget or n >= max_steps:
                break
        return (time.perf_counter() - t0) / n, n, False

    main_scene = scene(n_rays)
    dt, n, warm_only = timed(main_scene, n_samples, "train", budget_s)
    note = ("DIFFERENT WORKLOAD from the GPU line: %d rays x %d samples, HASH grid L=16 T=2^19 (the reference's CPU-runnable "
            "grids/hash_grid_torch.py, restated op for op in oracle/), density+colour decoders only (no panoptic heads, no delta grid), "
            "compositing, rgb L1, forward+backward+Adam, torch-CPU, %d of %d host threads; %s" %
            (n_rays, n_samples, torch.get_num_threads(), cores_available,
             ("1 step of %.1f s (warm-up only; host too slow for more)" % dt) if warm_only else ("%d timed steps of %.2f s" % (n, dt))))
    pts = []
    per = point_budget_s / max(1, 3 * len(points))
    for nr, ns in points:
        sc = scene(nr)
        for mode in ("encode", "forward", "train"):
            d, k, cold = timed(sc, ns, mode, per, max_steps=20)
            pts.append(dict(rays=nr, samples_per_ray=ns, mode={"encode": "encode forward", "forward": "encode + decoders + compositing forward",
                                                               "train": "train step (forward + backward + Adam)"}[mode],
                            s_per_step=round(d, 4), rays_s=round(nr / d, 1), samples_s=round(nr * ns / d, 1), steps=k, first_call_only=cold))
    return dict(value=n_rays / dt, unit="rays/s", cores=torch.get_num_threads(), cores_available=cores_available, kind="port", sample=note,
                threads_rationale=("torch-CPU on this workload's tensor shapes stops scaling at a few dozen threads: the per-level gather / index_add "
                                   "ops of grids/hash_grid_torch.py are launched as hundreds of small parallel regions per step, and with all 256 "
                                   "hardware threads of the GPU box's host the same step ran ~400x slower than with 8 (oversubscribed OpenMP barriers); "
                                   "min(32, cores_available) is the fastest setting measured there, so BASELINE.md section 3's 'all cores' is reported as "
                                   "cores_available next to the cores actually used"),
                torch=torch.__version__, points=pts)


# -------------------------------------------------------------------------- algorithmic bytes / flops per entry point
DECODER_PARAMS_PANOPTIC = (48 * 64 + 64) + (64 * 6 + 6) + (48 * 64 + 64) + (64 * 64 + 64) + (64 * 200 + 200)      # sem 48-64-6 + inst 48-64-64-200, with biases
DECODER_MACS = dict(density=48 * 64 + 64 * 16, colour=43 * 64 + 64 * 64 + 64 * 3, sem=48 * 64 + 64 * 6, inst=48 * 64 + 64 * 64 + 64 * 200)


def algorithmic_model(grid, M, N, channels, L_, F_, verts, bf16, head_once=None):
    """Per C-ABI entry point and STEP: dict(bytes=algorithmic HBM bytes, flops=useful matrix-core FLOPs, parts={launch: bytes per sample}).
    Bytes = every tensor a launch of the PRODUCTION path (bf16 features in the XCD8 layout, fused backward kernels: DESIGN 4.3b / 4.4b)
    must read or write once; table rows count as gathered bytes (rows x F x 4).  The per-sample figures are the table of DESIGN section 5
    (pinned by tests/test_abi_and_host.py::test_bench_byte_model_matches_design_table) and are what `kernels.*.pmc_bytes` - the rocprofv3
    FETCH_SIZE / WRITE_SIZE passes committed under profiles/ - is compared with in the bench line."""
    s = 2 if bf16 else 4
    C = L_ * F_
    feat = 128 if bf16 else C * 4               # XCD8: 8 groups x 16-byte piece per sample (zero padded), else the fp32 [M, C] row
    gather = L_ * verts * F_ * 4
    pan = "semantics" in channels
    mac = dict(DECODER_MACS, density=C * 64 + 64 * 16, sem=C * 64 + 64 * 6, inst=C * 64 + 64 * 64 + 64 * 200)
    used = ["density", "colour"] + (["sem", "inst"] if pan else [])
    macs_fwd = sum(mac[k] for k in used)
    # backward: dX = W^T dZ of every layer whose input gradient is needed (the colour decoder's first layer only towards its 16
    # density features) + dW = dZ^T A of every layer; the hidden activations the fused kernels recompute are not counted as useful
    macs_bwd_data = dict(density=mac["density"], colour=64 * 3 + 64 * 64 + 16 * 64, sem=mac["sem"], inst=mac["inst"])
    macs_bwd = sum(macs_bwd_data[k] + mac[k] for k in used)
    fwd_parts = dict(density=feat + 16 * s,                                   # features in, [M,16] density features out
                     colour=16 * s + 4 + 12 + 4)                               # x1, per-sample ray index, rgb f32, sigma f32 (view embedding: per ray)
    bwd_parts = dict(density=feat + 16 * s + feat,                             # features (recompute), upstream gradient, d features
                     colour=16 * s + 4 + 12 + 4 + 4 + 16 * s)                  # x1, ray index, d rgb, d sigma, sigma gate, d x1 (per-wave weight-gradient slabs: 11 - 45 MB per launch, not per sample)
    if head_once is None:
        head_once = M >= 160 * N        # pagnerf_amd.ops.HEAD_FWD_ONCE_MIN_PER_RAY: long rays take the 200-way head's one-launch forward
    if pan and head_once:
        # features, last hidden layer out, softmax statistics, semantic probabilities, compositing weight (+ the [N,200] per-ray sums, below):
        # decoder and per-ray weighted sum in ONE launch (pag_mlp_fwd_args.composite) - no pag_head_composite_fwd call
        fwd_parts["inst_once+sem"] = feat + 64 * s + 8 + 6 * s + 4
    elif pan:
        fwd_parts["inst_stats+sem"] = feat + 64 * s + 8 + 6 * s                # features, last hidden layer out, softmax statistics, semantic probabilities
    if pan:
        bwd_parts["inst_stage_A"] = 64 * s + 8 + 64 * s                        # hidden layer, statistics, hidden gradient out (rank-1 upstream gradient: per ray)
        bwd_parts["inst_stage_B+sem"] = feat + 64 * s + 6 * s + feat           # features, hidden gradient in, semantic probabilities, summed d features
    out = {
        "pag_mlp_fwd": dict(bytes=M * sum(fwd_parts.values()) + (N * 200 * 4 if pan and head_once else 0), flops=2 * M * macs_fwd, parts=fwd_parts),
        "pag_mlp_bwd": dict(bytes=M * sum(bwd_parts.values()), flops=2 * M * macs_bwd, parts=bwd_parts),
        "pag_composite_fwd": dict(bytes=M * (4 + 4 + 4 + 12 + 4) + N * 24, flops=0),
        "pag_composite_bwd": dict(bytes=M * (4 + 4 + 4 + 12 + 4 + 4 + 12) + N * 24, flops=0),
    }
    # the encoders: SURVEY 8d accounting.  forward = xyz + gathered rows + the feature row written (C x s useful bytes of the 128-byte piece);
    # `_add` also reads the other grid's row; backward = xyz + gradient row + read-modify-write of the gathered rows (the binned kernels
    # replace the RMW by a sort through a workspace: their minimum is xyz + gradient row + the table written once, `bytes_min`)
    out["pag_%s_encode_fwd" % grid] = dict(bytes=(12 + gather + C * s) * M, flops=0)
    out["pag_%s_encode_fwd_add" % grid] = dict(bytes=(12 + gather + 2 * C * s) * M, flops=0)
    rows = (1 << 18) if grid == "permuto" else (1 << 19)
    out["pag_%s_encode_bwd_set" % grid] = dict(bytes=(12 + C * s + 2 * gather) * M * (2 if pan else 1), flops=0,
                                               bytes_min=((12 + C * s) * M + L_ * rows * F_ * 4) * (2 if pan else 1))
    # the optimiser: p, g, m, v read and p, m, v written per fp32 parameter - the two tables (rows x F per level) and the 35 k decoder weights;
    # independent of the batch
    # the delta grid only receives a gradient through the panoptic heads (panoptic_delta_nef.py:219-226: its features feed sem / inst only), and a
    # parameter without a gradient is not stepped (torch.optim.Adam skips it; pagnerf_amd.optim.Adam issues no launch for it): one table in the rgb /
    # rgb + depth regimes, two with the panoptic channels on
    n_par = (2 if pan else 1) * L_ * rows * F_ + (35169 if pan else 35169 - DECODER_PARAMS_PANOPTIC)
    out["pag_adam_step"] = dict(bytes=28 * n_par, flops=0)
    if pan:
        if not head_once:
            out["pag_head_composite_fwd"] = dict(bytes=M * (64 * s + 8 + 4) + N * 200 * 4, flops=0, rebuild_flops=2 * M * 64 * 200)
        out["pag_composite_feats_fwd"] = dict(bytes=M * (6 * s + 4) + N * 6 * 4, flops=0)
    # pose optimisation (pc_nerf/ba_pipeline.py:85-92): the main grid's position gradient = a second gather pass (xyz, gradient row, the
    # gathered rows again, d xyz out; its per-XCD partial sums - 96 B per sample written and read once - are scratch, not counted), then the
    # per-ray sums of d xyz and d xyz * depth (d xyz + depth in, 24 B per ray out)
    out["pag_%s_encode_bwd_xyz" % grid] = dict(bytes=(12 + C * s + gather + 12) * M, flops=0, scratch_bytes=2 * 8 * 12 * M)
    out["pag_ray_sample_grad"] = dict(bytes=M * (12 + 4) + N * 24, flops=0)
    # the same gather pass with the per-ray reduction inside it (ABI 11, what the tracer takes on marched samples): xyz, gradient row, the gathered rows
    # again, the sample's depth and ray id in, 24 B per RAY out; the per-(XCD group, wave, ray) slots - 6 floats, written and read once - are scratch
    out["pag_%s_encode_bwd_rays" % grid] = dict(bytes=(12 + C * s + gather + 4 + 4) * M + N * 24, flops=0, scratch_bytes=2 * 8 * 24 * (M // 64 + N))
    return out


def kernel_table(prof_all, n_steps, model, pmc_blob=None):
    """Per C-ABI entry point: calls and device ms per step (HIP events around every call of an eager pass), algorithmic bytes / FLOPs of
    algorithmic_model() and the fractions of the HBM roof / the dense bf16 MFMA peak they amount to.
    -> (table, device ms of the decoder entry points, their useful FLOPs)."""
    import numpy as np
    kernels, mfma_ms, mfma_flops = {}, 0.0, 0.0
    for k, v in sorted(prof_all.items()):
        ms = float(np.sum(v)) / n_steps
        ent = dict(calls_per_step=len(v) / n_steps, ms_per_step=round(ms, 4))
        if k in model and ms > 0:
            md = model[k]
            by, fl = md["bytes"], md["flops"]
            ent.update(algorithmic_bytes=int(by), hbm_frac=round(by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
            if "bytes_min" in md:
                ent["algorithmic_bytes_min"] = int(md["bytes_min"])
            if "scratch_bytes" in md:
                ent["scratch_bytes"] = int(md["scratch_bytes"])
            if "parts" in md:
                ent["bytes_per_sample"] = md["parts"]
            if pmc_blob is not None:
                pb = pmc_bytes_per_step(pmc_blob, k, ent["calls_per_step"])
                if pb:
                    ent.update(pmc_bytes=pb, pmc_hbm_frac=round(pb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
            if fl:
                ent.update(mfma_tflops=round(fl / (ms * 1e-3) / 1e12, 2), mfma_frac=round(fl / (ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 5))
                mfma_ms += ms
                mfma_flops += fl
            if md.get("rebuild_flops"):
                mfma_ms += ms          # the head's probabilities are rebuilt on the matrix cores: device time of the decoders, no useful FLOPs credited
        kernels[k.replace("pag_", "")] = ent
    return kernels, mfma_ms, mfma_flops


# C-ABI entry point -> substrings of the kernel names rocprofv3 reports for it (profiles/*_pmc_traffic_per_launch.json keys)
PMC_KERNELS = {
    "pag_permuto_encode_fwd": ["permuto_fwd_kernel"], "pag_permuto_encode_fwd_add": ["permuto_fwd_add_kernel"],
    "pag_permuto_encode_bwd_set": ["bin_kernel", "reduce_kernel"],
    "pag_mlp_fwd": ["mlp_fwd_fast<2, 0", "mlp_fwd_fast<3, 1", "mlp_fwd_density_colour", "mlp_fwd_wide_stats", "head_fwd_once_kernel"],
    "pag_mlp_bwd": ["mlp_bwd_fused<2, 0", "mlp_bwd_fused<3, 1", "mlp_bwd_wide_blocks", "mlp_bwd_pair", "wgrad_finish_kernel"],
    "pag_head_composite_fwd": ["head_composite_fwd_kernel"], "pag_composite_fwd": ["composite_fwd_kernel"],
    "pag_composite_bwd": ["composite_bwd_kernel"], "pag_composite_feats_fwd": ["composite_feats_small_fwd_kernel"],
}


def pmc_bytes_per_step(blob, entry, calls_per_step):
    """HBM bytes per STEP of one entry point from a committed per-launch PMC file: the sum over its kernels of bytes per launch x the
    launches of that kernel per call (1) x calls per step of the kernels that belong to ONE call each (encode backward: 2 grids)."""
    names = PMC_KERNELS.get(entry)
    if not names:
        return None
    total, found = 0.0, 0
    for sub in names:
        for k, v in blob.items():
            if isinstance(v, dict) and sub in k and "hbm_bytes_per_launch_corrected" in v and not (sub == "permuto_fwd_kernel" and "fwd_add" in k):
                total += v["hbm_bytes_per_launch_corrected"]
                found += 1
                break
    if not found:
        return None
    if entry.endswith("encode_bwd_set") or entry.endswith("encode_fwd") or entry.endswith("encode_fwd_add"):
        total *= calls_per_step           # one bin + one reduce launch (or one encode launch) per call
    # pag_adam_step: one launch per parameter group (both tables in the first; the decoders' launch is the same kernel name, 5 us: the mean of the
    # committed per-launch figure mixes the two - read the optimiser's traffic from its algorithmic bytes instead)
    return int(total)


# ------------------------------------------------------------------------------- the roof that binds the encode launch: row requests
def request_rate(nef, rays, tracer, enc_ms, table_dtype):
    """What limits the grid-interpolate launch is not bytes but the RATE OF ROW REQUESTS THAT MISS THE L1 (profiles/README.md
    round 3: ~0.4 lane-requests per clock and CU; a request costs the same whether it returns 4, 8 or 16 bytes - which is why a byte
    fraction reads low for fp16 tables).  Measured live: the same launch on the same samples with EVERY level at the finest scale /
    resolution, where no two lanes share a row and every one of the L x V gathers per sample misses: `all_miss_rows_per_s` is the
    request ceiling of this chip for this sample set.  `rows_per_s` is the launch as benched; `frac_of_all_miss_time` = its time over
    the all-miss time: how much of the request-bound worst case the locality of the coarse levels (L1 hits between neighbouring samples
    of a ray) and the per-XCD level tables (L2 hits) take off - the part that is left is the fine levels' requests, at the ceiling."""
    import numpy as np
    import torch
    from pagnerf_amd import ops
    g = nef.grid
    with torch.no_grad():
        out = g.raymarch(rays() if callable(rays) else rays, level=None, num_samples=tracer.num_steps, raymarch_type=tracer.raymarch_type)
        xyz = out[2].reshape(-1, 3).contiguous()
        M = xyz.shape[0]
        Lv, Fd = g.tables.shape[0], g.tables.shape[2]
        if hasattr(g, "random_shift_per_level"):
            verts = 4
            finest = float(np.min(np.asarray(g.resolutions)))
            spec = ops.permuto_spec(type(g).scale_factors(np.asarray([finest] * Lv)), g.random_shift_per_level, g.tables.shape[1], Fd,
                                    half_coords=bool(g.half_coords))
        else:
            verts = 8
            finest = float(np.max(np.asarray(g.resolutions)))
            spec = ops.hash_spec([finest] * Lv, g.codebook_bitwidth, Fd, half_coords=bool(g.half_coords))
        lay = "xcd8" if nef._grouped() else None
        fn = lambda: ops.encode(xyz, g.tables.detach(), spec, None, nef.feat_dtype, layout=lay)
        for _ in range(2):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(5):
            fn()
        b.record()
        torch.cuda.synchronize()
        miss_ms = a.elapsed_time(b) / 5
    rows = M * Lv * verts
    ent = dict(rows_per_launch=rows, rows_per_sample=Lv * verts, all_miss_launch_ms=round(miss_ms, 4),
               all_miss_rows_per_s=round(rows / (miss_ms * 1e-3), 1), table_dtype=table_dtype,
               note="every level at the finest scale on the same samples: every gather misses the L1 - the request ceiling, measured in this run")
    if enc_ms:
        ent.update(rows_per_s=round(rows / (enc_ms * 1e-3), 1), frac_of_all_miss_time=round(enc_ms / miss_ms, 4))
    return ent


# ------------------------------------------------------------------------------------------- forward-only render (validation path)
def render_image_line(args, dev, all_ch, out_bytes, H=720, W=1280, render_batch=8000, images=2, pruned=False):
    """The reference's OTHER caller of the path: pc_nerf/trainer.py:943-999 validate -> :637-649 batch_render - one H x W image
    (BUP20: 720 x 1280) through pagnerf_amd.batch_render under torch.no_grad(), render_batch = 8000 rays x 512 samples
    (best.yaml:143,146), dense occupancy (every sample survives: the worst case for the renderer).  Reported for all channels and for
    rgb + depth: ms per image, rays/s, per C-ABI entry point ms per image (HIP events around every call in a separate pass), and the
    encode kernel's fraction of the HBM roof (876 B per sample and grid) over its launches in that pass."""
    import numpy as np
    import torch
    import pagnerf_amd
    from pagnerf_amd import ops
    nef = make_model(args, dev, seed=0).eval()
    n = H * W
    rays, _ = make_rays(n, dev, seed=77)
    if pruned:
        # what every validation after epoch 201 runs (pc_nerf/trainer.py:362-366 switches nef and tracer to the voxel march with samples_per_voxel
        # steps when voxel_raymarch_epoch_start = 201 is reached, best.yaml:34,31; validate() -> batch_render, :637-649, uses the tracer as it stands)
        occ = synthetic_prune(nef, args.occupancy)
        tracer = pagnerf_amd.PanopticPackedRFTracer(raymarch_type="voxel", num_steps=2, bg_color="white", ray_max_travel=6.0, use_graphs=False)
        with torch.no_grad():
            warm = pagnerf_amd.Rays(rays.origins[:render_batch], rays.dirs[:render_batch], rays.dist_min, rays.dist_max)
            mo = nef.grid.raymarch(warm, level=None, num_samples=2, raymarch_type="voxel", max_travel=6.0)
        out = dict(image="%d x %d = %d rays, voxel march (2 samples per voxel) on the pruned grid (%.1f %% of 128^3 cells occupied: ~%d samples per ray), render_batch %d "
                         "(%d traces per image), torch.no_grad()" % (H, W, n, 100 * occ, int(mo[2].shape[0] * 2 / render_batch), render_batch,
                                                                     (n + render_batch - 1) // render_batch))
    else:
        tracer = pagnerf_amd.PanopticPackedRFTracer(raymarch_type="ray", num_steps=512, bg_color="white", use_graphs=False)
        out = dict(image="%d x %d = %d rays x 512 samples, render_batch %d (%d traces per image), dense occupancy, torch.no_grad()"
                         % (H, W, n, render_batch, (n + render_batch - 1) // render_batch))
    pipe = pagnerf_amd.Pipeline(nef, tracer)
    bps = 12 + 24 * 4 * 2 * (2 if args.table_dtype == "fp16" else 4) + 24 * 2 * out_bytes
    with torch.no_grad():
        big = 32768       # the same image with a render batch sized for this GPU's memory instead of the reference's 8000 (`render_batch` is a config key)
        for tag, chans, render_batch in (("all_channels", sorted(all_ch), render_batch), ("rgb_depth", ["depth", "rgb"], render_batch),
                                         ("all_channels_render_batch_%d" % big, sorted(all_ch), big)):
            warm = pagnerf_amd.Rays(rays.origins[:4 * render_batch], rays.dirs[:4 * render_batch], rays.dist_min, rays.dist_max)
            pagnerf_amd.batch_render(pipe, warm, channels=chans, render_batch=render_batch)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(images):
                rb = pagnerf_amd.batch_render(pipe, rays, channels=chans, render_batch=render_batch)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / images
            assert rb.rgb.shape[0] == n
            ops.profile_start()
            pagnerf_amd.batch_render(pipe, rays, channels=chans, render_batch=render_batch)
            prof = ops.profile_stop()
            per = {k.replace("pag_", ""): round(float(np.sum(v)), 3) for k, v in sorted(prof.items()) if float(np.sum(v)) > 0.05}
            enc = prof.get("pag_%s_encode_fwd" % args.grid, [])
            ent = dict(channels=chans, render_batch=render_batch, ms_per_image=round(dt * 1e3, 2), rays_s=round(n / dt, 1),
                       entry_points_ms_per_image=per, device_ms_per_image=round(float(sum(np.sum(v) for v in prof.values())), 2))
            if not pruned:
                ent["samples_s"] = round(n * 512 / dt, 1)
            if enc and not pruned:
                m_launch = render_batch * 512
                full = [e for e in enc][:n // render_batch]          # the full-size launches (the last chunk of an image is shorter)
                ent["encode_fwd"] = dict(launches=len(enc), avg_launch_ms=round(float(np.mean(full)), 4),
                                         hbm_frac=round(bps * m_launch / (float(np.mean(full)) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                         bytes_per_sample=bps)
            out[tag] = ent
    del nef, pipe, rays
    torch.cuda.empty_cache()
    return out


# -------------------------------------------------------------------------------------------------- dry run (CPU, gloo)
def dry_run_rank(args, world, rank):
    """No kernels: the process group, shard.GradSync / all_gather_render on CPU tensors, the timing protocol and the JSON line."""
    import torch
    if args.sparse_sync == "off":
        args.sparse_sync = "bounded"
    import torch.distributed as dist
    from pagnerf_amd import shard, RenderBuffer
    if world > 1:
        dist.init_process_group("gloo")
        assert dist.get_world_size() == args.gpus, "world size %d != --gpus %d" % (dist.get_world_size(), args.gpus)
    seen = torch.ones(1)
    if world > 1:
        dist.all_reduce(seen)
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.zeros(64, 8)), torch.nn.Parameter(torch.zeros(16))]
    comm = {"bf16": torch.bfloat16, "fp32": None, "auto": "auto"}[args.grad_sync]
    sync = shard.GradSync(params, early=[params[0]], comm_dtype=comm, big=256) if world > 1 else None
    n_local = 8

    def step():
        for p in params:
            p.grad = None
        ((params[0] * (rank + 1)).sum() + (params[1] * 2).sum()).backward()
        if sync is not None:
            sync.finish()
    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        mean = sum(r + 1 for r in range(world)) / world
        assert torch.allclose(params[0].grad, torch.full((64, 8), mean), rtol=2.0 ** -7 if comm is not None else 1e-6), \
            "GradSync over the bench's process group gave a wrong mean"
        # the weak-scaling regime lines of the real run, as plumbing: three "regimes" whose step time is a sleep, each with its own GradSync(comm_dtype=
        # "auto") over a table-sized gradient that is NOT early (the exposed exchange); the assumed bus bandwidth is set so that the predicted fp32
        # exchange is 15 ms - the 120 ms "dense" step keeps fp32, the short "post-prune" steps (under 60 ms) switch to the bf16 direct reduce
        # Each regime also runs the touched-rows exchange (GradSync(sparse="bounded"), shard.SparseRows) on a table-shaped gradient [levels, rows, features]
        # whose rows are zero outside a regime-dependent set: every row in the dense regime, the coarse half of the levels at 5 % after the "prune" -
        # `exchanged_bytes` is what the last step moved per rank and table, `dense_bytes` what the whole table would have
        weak = []
        Lt, Tt, Ft = 4, 4096, 2
        tab_bytes = Lt * Tt * Ft * 4
        bus = 2.0 * (world - 1) / world * tab_bytes / 15e-3 / 1e9       # predicted fp32 exchange: 15 ms (gloo on CPU tensors: a step here is ~25 ms of host work)
        gen = torch.Generator().manual_seed(5)                         # the same masks on every rank + a rank-dependent part
        for name, sleep_ms, fill in (("weak_dense_all_channels", 120.0, (1.0, 1.0, 1.0, 1.0)), ("weak_post_prune_rgb", 0.3, (0.05, 0.05, 1.0, 1.0)),
                                     ("weak_post_prune_all_channels", 0.6, (0.05, 0.05, 1.0, 1.0))):
            ps = [torch.nn.Parameter(torch.zeros(Lt, Tt, Ft)), torch.nn.Parameter(torch.zeros(16))]
            keep = (torch.rand(Lt, Tt, 1, generator=gen) < torch.tensor(fill)[:, None, None]).float()
            sy = shard.GradSync(ps, comm_dtype="auto", big=256, bus_gbs=bus, sparse=(False if args.sparse_sync == "off" else args.sparse_sync))
            t1 = time.perf_counter()
            n_it = shard.AUTO_WARM + 4
            for _ in range(n_it):
                for q_ in ps:
                    q_.grad = None
                ((ps[0] * keep * (rank + 1)).sum() + (ps[1] * 2).sum()).backward()
                time.sleep(sleep_ms * 1e-3)
                sy.finish()
            ent = dict(name=name, ms_per_step=round((time.perf_counter() - t1) / n_it * 1e3, 3), grad_sync=sy.auto_decision)
            st = sy.sparse_stats()
            if st:
                ent.update(sparse_sync=args.sparse_sync, exchanged_bytes=st[0]["exchanged_bytes"], dense_bytes=st[0]["dense_bytes"], bitmap_bytes=st[0]["bitmap_bytes"],
                           whole_levels=st[0]["whole_levels"], dropped_rows=st[0]["dropped_rows"])
            weak.append(ent)
            assert torch.allclose(ps[0].grad, keep.expand(Lt, Tt, Ft) * mean, rtol=2.0 ** -7)
            sy.remove()
        lo, hi = shard.shard_bounds(n_local * world, rank, world)
        rb = shard.all_gather_render(RenderBuffer(rgb=torch.arange(lo, hi, dtype=torch.float32)[:, None].repeat(1, 3)), n_local * world)
        assert torch.equal(rb.rgb[:, 0], torch.arange(n_local * world, dtype=torch.float32))
    if rank == 0:
        emit(dict(metric="rays/sec (train step) on BUP20-shape scene", value=0.0, unit="rays/s", n_gpus=world,
                  steps=args.steps, warmup=args.warmup, ms_per_step=round(dt / max(args.steps, 1) * 1e3, 3),
                  higher_is_better=True, scaling="weak", vs_baseline=None, dtype=args.precision, data="synthetic",
                  config=dict(workload="DRY RUN: no kernels, gloo on CPU tensors - plumbing check only"),
                  dry_run=True, rccl_ranks_seen=int(seen.item()), backend="gloo", grad_sync=args.grad_sync,
                  weak_regimes=(weak if world > 1 else None)))
    if world > 1:
        dist.destroy_process_group()
    return 0


# ---------------------------------------------------------------------------------------------------------- one rank
def run_rank(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE %d - refusing to report a run of the wrong size" % (args.gpus, world), file=sys.stderr)
        return 2
    if args.dry_run:
        return dry_run_rank(args, world, rank)
    import numpy as np
    import torch
    import torch.distributed as dist
    # PAG_BENCH_SHARE_GPU=1 (testing only): every rank uses cuda:0 and the collectives go through gloo, so the multi-process
    # code path can be exercised on a one-GPU box; the real launch is one rank per GPU over RCCL ("nccl" on ROCm).
    share = os.environ.get("PAG_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    backend = None
    if world > 1:
        backend = "gloo" if share else "nccl"
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        assert dist.get_world_size() == args.gpus
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    ranks_seen = 1
    if world > 1:
        dist.barrier()
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)                                     # every rank of the process group contributes 1
        ranks_seen = int(ones.item())
        if ranks_seen != args.gpus:
            print("bench.py: all-reduce saw %d ranks, expected %d" % (ranks_seen, args.gpus), file=sys.stderr)
            return 3
    from pagnerf_amd import ops, shard

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(dt):
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    class Job:
        """One configuration: model, tracer, rays, optimiser, gradient sync."""

        def __init__(self, rays_n, samples, grid, channels, raymarch="ray", pose=False, total_rays=None, seed=0, sparse=None, **mk):
            self.nef = make_model(args, dev, seed=seed, grid=grid, **mk)          # same seed everywhere: replicated parameters
            self.tracer = make_tracer(args, raymarch, samples)
            self.occupied = synthetic_prune(self.nef, args.occupancy) if raymarch == "voxel" else 1.0
            self.channels = channels
            extra = []
            if pose:
                total = total_rays or rays_n
                lo, hi = shard.shard_bounds(total, rank, world) if total_rays else (0, rays_n)
                self.pose = PoseOpt(self.nef, self.tracer, total, args.images, dev, lo, hi)
                _, self.gt = make_rays(hi - lo, dev, seed=1000 + rank)
                self.rays = self.pose.rays
                extra = self.pose.parameters()
            else:
                self.rays, self.gt = make_rays(rays_n, dev, seed=1000 + rank)      # per-rank ray shard
            self.opt = make_optimizer(self.nef, extra)
            self.lin_assign = None
            self.images, self.points_fn, self.seg_reg, self.overlap = 1, None, False, False
            self.sync = None
            if world > 1:
                early = [self.nef.delta_grid.tables] if hasattr(self.nef, "delta_grid") else []
                self.sync = shard.GradSync(list(self.nef.parameters()) + list(extra), early=early,
                                           comm_dtype={"bf16": torch.bfloat16, "fp32": None, "auto": "auto"}[args.grad_sync],
                                           sparse=(sparse if sparse is not None else (False if args.sparse_sync == "off" else args.sparse_sync)))

        def step(self, channels=None):
            return train_step(self.nef, self.tracer, self.opt, self.rays, self.gt, channels or self.channels, world, self.sync,
                              lin_assign=self.lin_assign, images=self.images, points_fn=self.points_fn, seg_reg=self.seg_reg, overlap=self.overlap)

        def timed(self, n_steps, channels=None, profile=None):
            """profile (a set of C-ABI entry points): HIP events around those calls - which only exist on the EAGER path (a graph replay
            issues no per-kernel host call), so a profiled region runs with the tracer's graphs switched off."""
            was = self.tracer.use_graphs
            if profile is not None:
                self.tracer.use_graphs = False
            # the interpreter's cyclic collector is not part of a step: collected before, kept from firing inside the timed region (one full
            # collection is a ~10 ms host stall = +0.5 ms on each of 20 steps - seen once in a dozen runs), back on afterwards
            import gc
            gc.collect()
            gc_was = gc.isenabled()
            gc.disable()
            # the collection above leaves the GPU idle for ~10 ms and the chip answers with a few slow steps (scripts/step_series.py: 3.73, 3.61, 3.56, 3.55 ms
            # after such a pause against 3.43 in steady state): a handful of UNTIMED steps bring it back to the state every other step of a training run sees
            # before the barrier + synchronize that open the timed region (reported as config.settle_steps)
            for _ in range(SETTLE_STEPS if profile is None else 0):
                self.step(channels)
            barrier()
            if profile is not None:
                ops.profile_start(only=profile)
            t0 = time.perf_counter()
            for _ in range(n_steps):
                self.step(channels)
            barrier()
            dt = time.perf_counter() - t0
            if gc_was:
                gc.enable()
            prof = ops.profile_stop() if profile is not None else None
            self.tracer.use_graphs = was
            return max_over_ranks(dt), prof

        def graph_stats(self):
            g = getattr(self.tracer, "_graphs", None)
            return None if g is None else dict(captures=g.captures, replays=g.replays, overflows=g.overflows)

        def samples_per_step(self):
            with torch.no_grad():
                r = self.rays() if callable(self.rays) else self.rays
                out = self.nef.grid.raymarch(r, level=None, num_samples=self.tracer.num_steps, raymarch_type=self.tracer.raymarch_type)
            return int(out[2].shape[0] * (out[2].shape[1] if out[2].dim() == 3 else 1))

        def close(self):
            if self.sync is not None:
                self.sync.remove()

    all_ch = {"rgb", "depth", "semantics", "inst_embedding"}
    channels = all_ch if args.channels == "all" else ({"rgb", "depth"} if args.channels == "rgbd" else {"rgb"})
    job = Job(args.rays, args.samples, args.grid, channels, raymarch=args.raymarch, pose=args.pose_opt)
    if args.lin_assign and args.pose_opt and args.channels == "all":
        from pagnerf_amd.loss import LinAssignmentThingsLoss
        job.lin_assign, job.images, job.points_fn, job.seg_reg = LinAssignmentThingsLoss(outlier_rejection=True), args.images, job.pose.points_3d, True
        if args.two_call:
            job.overlap, job.tracer.graph_split = True, True
    enc_name = "pag_%s_encode_fwd" % args.grid

    for _ in range(args.warmup):
        job.step()
    # Graph capture is one-time set-up (an eager step that observes the sample count, then the capture): with fewer warm-up steps than that
    # takes it would land inside the K timed steps.  Extra UNTIMED steps until a graph has replayed; reported as graphs.priming_steps.
    priming = 0
    if job.tracer.use_graphs:      # (every rank takes the same number of steps: the count is a function of the step index)
        while priming < 4 and not ((job.graph_stats() or {}).get("replays", 0) > 0):
            job.step()
            priming += 1
    # The K timed steps.  With graphs on (default at N = 1) a step issues the ray march and two graph replays, so no per-kernel host
    # call exists to bracket with events: the roofline kernel's duration is then measured over K further EAGER steps right after the
    # timed region (same process, same inputs, HIP events on the launch stream around its C-ABI call).  With --graphs off the events
    # sit inside the timed region itself, as in rounds 1 - 2.
    graphs_on = bool(job.tracer.use_graphs)       # pose-optimisation steps replay graphs too (d origins / d dirs are outputs of the backward graph)
    if graphs_on:
        dt, _ = job.timed(args.steps)
        _, prof = job.timed(args.steps, profile={enc_name})
    else:
        dt, prof = job.timed(args.steps, profile=None if os.environ.get("PAG_BENCH_PROFILE_ALL") else {enc_name})

    M = job.samples_per_step() if (args.raymarch == "voxel" or args.pose_opt) else args.rays * args.samples
    L_, F_ = (24, 2) if args.grid == "permuto" else (16, 2)
    verts = 4 if args.grid == "permuto" else 8
    out_bytes = 2 if args.precision == "bf16" else 4
    table_bytes = 2 if args.table_dtype == "fp16" else 4
    bytes_per_sample = 12 + L_ * verts * F_ * table_bytes + L_ * F_ * out_bytes       # xyz + table gathers + feature row
    # HBM bytes per launch of that kernel from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected separately, KB
    # units, FETCH_SIZE x2 on gfx950 as MI355X_MICROARCH.md prescribes) - measured offline on this exact configuration and
    # committed under profiles/ (file name and the commit it was taken at are stamped next to the number); null otherwise.
    traffic = traffic_src = None
    pdir = os.path.join(ROOT, "profiles")
    cands = sorted(f for f in os.listdir(pdir) if f.endswith("_pmc_traffic_per_launch.json"))
    if cands and (args.grid, args.rays, args.samples, args.precision, args.raymarch) == ("permuto", 4096, 512, "bf16", "ray"):
        blob = json.load(open(os.path.join(pdir, cands[-1])))      # newest committed PMC pass (profiles/README.md)
        for k, v in blob.items():
            if isinstance(v, dict) and "permuto_fwd_kernel" in k and "permuto_fwd_add_kernel" not in k and "hbm_bytes_per_launch_corrected" in v:
                traffic = v["hbm_bytes_per_launch_corrected"]
        traffic_src = dict(file="profiles/" + cands[-1], commit=blob.get("_commit"), note="offline rocprofv3 --pmc passes, not this run")
    enc_ms = (prof or {}).get(enc_name, [])
    roofline = None
    if enc_ms:
        mean_ms = float(np.mean(enc_ms))
        achieved = bytes_per_sample * M / (mean_ms * 1e-3) / 1e9
        roofline = dict(bound="hbm", kernel=enc_name.replace("pag_", "") + "_kernel", achieved=round(achieved, 1),
                        peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic,
                        traffic_source=traffic_src, launches=len(enc_ms), avg_launch_ms=round(mean_ms, 4),
                        algorithmic_bytes_per_launch=bytes_per_sample * M,
                        timing=("HIP events around the kernel's C-ABI call in %d eager steps run right after the timed region (the timed steps "
                                "replay HIP graphs: no per-launch host call to bracket)" % args.steps) if graphs_on else
                               "HIP events around the kernel's C-ABI call inside the timed steps")

    if roofline is not None and not args.no_aux:
        roofline["request_rate"] = request_rate(job.nef, job.rays, job.tracer, roofline["avg_launch_ms"], args.table_dtype)
    line = dict(
        metric="rays/sec (train step) on BUP20-shape scene", value=round(world * args.rays * args.steps / dt, 1), unit="rays/s",
        n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(dt / args.steps * 1e3, 3),
        higher_is_better=True, scaling="weak", vs_baseline=None, dtype=args.precision, data="synthetic",
        config=dict(workload="BUP20-shaped single view, PanopticDeltaNeF + %s grid (main+delta), %d rays x %s per GPU "
                             "(M=%d packed samples), channels %s, train step fwd+bwd+Adam%s%s" %
                             ("permutohedral L=24 F=2 T=2^18" if args.grid == "permuto" else "hash L=16 F=2 T=2^19", args.rays,
                              ("%d samples" % args.samples) if args.raymarch == "ray" else
                              ("voxel march, %.1f %% of 128^3 cells occupied" % (100 * job.occupied)),
                              M, "+".join(sorted(channels)), ", pose-opt" if args.pose_opt else "",
                              ", RCCL grad all-reduce" if world > 1 else ""),
                    rays_per_gpu=args.rays, samples_per_ray=args.samples, grid=args.grid, channels=sorted(channels),
                    raymarch=args.raymarch, half_coords=(args.grid == "permuto" and not args.fp32_coords), table_dtype=args.table_dtype,
                    hip_graphs=bool(graphs_on and job.tracer.use_graphs is True), static_buffers=bool(job.tracer.use_graphs == "static"),
                    settle_steps=SETTLE_STEPS,
                    parallelism="ray-sharded data parallel x%d" % world),
        rccl_ranks_seen=ranks_seen, backend=backend, grad_sync=((job.sync.auto_decision or args.grad_sync) if world > 1 else None), roofline=roofline,
        graphs=(dict(job.graph_stats(), priming_steps=priming) if job.graph_stats() is not None else None))

    if not args.no_aux:
        # ---- per-entry-point device time: a separate, untimed pass with events around every C-ABI call
        n_bd = max(1, min(5, args.steps))
        barrier()
        was_graphs, job.tracer.use_graphs = job.tracer.use_graphs, False       # per-kernel events need the eager path
        ops.profile_start()
        for _ in range(n_bd):
            job.step()
        prof_all = ops.profile_stop()
        job.tracer.use_graphs = was_graphs
        model = algorithmic_model(args.grid, M, args.rays, channels, L_, F_, verts, args.precision == "bf16")
        pmc_blob = json.load(open(os.path.join(pdir, cands[-1]))) if (cands and traffic_src is not None) else None
        kernels, mfma_ms, mfma_flops = kernel_table(prof_all, n_bd, model, pmc_blob)
        line["kernels"] = kernels
        if pmc_blob is not None:
            line["kernels_pmc_source"] = traffic_src
        if mfma_ms:
            line["mfma_util"] = dict(
                algorithmic_tflops=round(mfma_flops / (mfma_ms * 1e-3) / 1e12, 2), peak_tflops=MFMA_PEAK_TFLOPS,
                frac=round(mfma_flops / (mfma_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 5), decoder_ms_per_step=round(mfma_ms, 4),
                useful_mac_per_sample=dict(fwd=int(model["pag_mlp_fwd"]["flops"] / (2 * M)), bwd_data_and_wgrad=int(model["pag_mlp_bwd"]["flops"] / (2 * M))),
                note="useful decoder MACs per sample (forward; backward = input gradients + every weight gradient, which pag_mlp_bwd forms in "
                     "the same launches) x 2 FLOP over the device time of pag_mlp_fwd + pag_mlp_bwd + pag_head_composite_fwd, against the dense "
                     "bf16 MFMA peak; recomputed activations / rebuilt probabilities are not counted as useful.  K <= 64 MLPs are bound by "
                     "activation traffic and per-tile VALU work (kernels.*.hbm_frac, DESIGN 4.3b / 4.4b), not by the matrix pipe")
        # ---- sustained: >= 2 s of the same step; the clocks / temperature of a 0.14 s burst are not what training sees
        if args.sustain_steps > 0:
            half = args.sustain_steps // 2
            d1, _ = job.timed(args.sustain_steps - half)
            d2, _ = job.timed(half)
            line["sustained"] = dict(steps=args.sustain_steps, seconds=round(d1 + d2, 3),
                                     ms_per_step=round((d1 + d2) / args.sustain_steps * 1e3, 3),
                                     ms_per_step_last_half=round(d2 / max(half, 1) * 1e3, 3),
                                     value_last_half=round(world * args.rays * half / d2, 1), unit="rays/s")
        # ---- the same step without graphs (what rounds 1 - 2 measured)
        if graphs_on:
            was_mode, job.tracer.use_graphs = job.tracer.use_graphs, False
            for _ in range(2):
                job.step()
            n_e = max(3, args.steps // 2)
            d_e, _ = job.timed(n_e)
            job.tracer.use_graphs = was_mode
            line["eager"] = dict(ms_per_step=round(d_e / n_e * 1e3, 3), rays_s=round(world * args.rays * n_e / d_e, 1),
                                 note="PanopticPackedRFTracer(use_graphs=False): every kernel launched from Python, host waits for the sample count")
        # ---- rgb-only regime (epochs < 601, best.yaml:89)
        if args.channels == "all":
            for _ in range(2):
                job.step({"rgb"})
            n_aux = max(3, args.steps // 2)
            dt_rgb, _ = job.timed(n_aux, {"rgb"})
            line["rgb_only"] = dict(workload="same scene, channels {rgb} only (epochs < 601, best.yaml:89)",
                                    value=round(world * args.rays * n_aux / dt_rgb, 1), unit="rays/s", ms_per_step=round(dt_rgb / n_aux * 1e3, 3))
        default_cfg = (args.grid, args.rays, args.samples, args.raymarch, args.pose_opt, args.channels) == ("permuto", 4096, 512, "ray", False, "all")
        if args.channels == "all" and world == 1:
            # ---- the late-training step as the trainer runs it: instance term = LinAssignmentThingsLoss on the rendered probabilities
            from pagnerf_amd.loss import LinAssignmentThingsLoss
            la = {}
            for tag, mod in (("device_solver", LinAssignmentThingsLoss()), ("device_solver_two_call", LinAssignmentThingsLoss()),
                             ("device_cost_matrix", LinAssignmentThingsLoss(solver="scipy")), ("device_cost_matrix_two_call", LinAssignmentThingsLoss(solver="scipy")),
                             ("reference_formulation", ReferenceFormulationThingsLoss())):
                job.lin_assign = mod
                two = tag.endswith("two_call")
                job.overlap, job.tracer.graph_split = two, (True if two else None)       # two backward graphs (rgb | panoptic), called in that order
                for _ in range(4 if two else 2):
                    job.step()
                n_la = max(3, args.steps // 2)
                d_la, _ = job.timed(n_la)
                la[tag] = dict(ms_per_step=round(d_la / n_la * 1e3, 3), rays_s=round(args.rays * n_la / d_la, 1))
            job.lin_assign, job.overlap, job.tracer.graph_split = None, False, None
            la["labels_per_image"] = int((torch.unique(job.gt["inst_ids"]) > 0).sum())
            la["note"] = ("same step with the instance term of trainer.py:483-520 (per-image Hungarian relabelling + NLL); device_solver = the default since round 6: "
                          "pag_assign_cost + pag_assign_solve (SciPy's algorithm on the device, same columns) + pag_assign_nll - no host wait in the step; device_cost_matrix = "
                          "pagnerf_amd.loss.LinAssignmentThingsLoss: pag_assign_cost (ids, sums, cost rows on the device), ONE copy + wait, SciPy, pag_assign_nll; "
                          "device_cost_matrix_two_call = the same with loss_fn.begin() / rgb_loss.backward() / loss_fn.finish() / rest.backward(): the colour / "
                          "density / main-grid backward runs while the host waits and solves the assignment; reference_formulation = one masked sum and one "
                          "device-to-host copy per label")
            line["with_lin_assignment"] = la
        job.close()
        del job
        torch.cuda.empty_cache()

        def short_run(name, n_steps, warm, key=None, **kw):
            j = Job(**kw)
            for _ in range(max(warm, 3)):                       # graphs: step 0 learns the sample count, step 1 captures
                j.step()
            d, _ = j.timed(n_steps)
            _, p = j.timed(min(n_steps, 5), profile={"pag_%s_encode_fwd" % kw["grid"]})      # eager pass: events around the encode launch
            m = j.samples_per_step() if (kw.get("raymarch") == "voxel" or kw.get("pose")) else kw["rays_n"] * kw["samples"]
            rays_total = kw.get("total_rays") or kw["rays_n"] * world
            ms = d / n_steps * 1e3
            e = p.get("pag_%s_encode_fwd" % kw["grid"], [])
            lv, vt = (24, 4) if kw["grid"] == "permuto" else (kw.get("num_lods") or 16, 8)
            tb = 2 if (kw.get("table_dtype") or args.table_dtype) == "fp16" else 4
            bps = 12 + lv * vt * 2 * tb + lv * 2 * out_bytes
            ent = dict(name=name, key=key, ms_per_step=round(ms, 3), rays_s=round(rays_total / ms * 1e3, 1), samples_per_step=int(m), steps=n_steps,
                       encode_bytes_per_sample=bps,
                       encode_frac=round(bps * m / (float(np.mean(e)) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if e else None)
            if kw.get("raymarch") == "voxel":
                ent["occupied_fraction"] = round(j.occupied, 4)
            if e:
                ent["request_rate"] = request_rate(j.nef, j.rays, j.tracer, float(np.mean(e)), kw.get("table_dtype") or args.table_dtype)
            ent["hip_graphs"] = j.graph_stats()
            if j.sync is not None:
                ent["grad_sync"] = j.sync.auto_decision or dict(comm_dtype="bf16" if j.sync.comm_dtype is not None else "fp32", decided="flag")
                st = j.sync.sparse_stats()
                if st:
                    ent["sparse_sync"] = [dict(exchanged_bytes=t["exchanged_bytes"], dense_bytes=t["dense_bytes"], whole_levels=t["whole_levels"],
                                               dropped_rows=t["dropped_rows"]) for t in st]
            j.close()
            del j
            torch.cuda.empty_cache()
            return ent

        def best_yaml_regimes(images=6, per_image=4096):
            from pagnerf_amd.loss import LinAssignmentThingsLoss
            total = images * per_image
            out, regimes = dict(rays_per_step=total, images=images, pose_optimisation=True), {}
            late = ("epochs 601 - 800: voxel march, all channels, LinAssignmentThingsLoss(outlier_rejection=True) on the 6-image batch + "
                    "segment_consistency_regularizer")
            specs = (("dense_rgb", "epochs 0 - 200: dense occupancy, 'ray' march x 512, channels rgb + depth", "ray", 512, {"rgb", "depth"}, 0, 6),
                     ("post_prune_rgb", "epochs 201 - 600: voxel march (2 samples per voxel), channels rgb + depth", "voxel", 2, {"rgb", "depth"}, 0, 20),
                     ("post_prune_all_assign_one_backward", late + "; ONE loss.backward() after the assignment (the reference's loop as it stands)", "voxel", 2,
                      set(all_ch), 1, 20),
                     ("post_prune_all_assign", late + "; two-call form (INTEGRATION.md): loss_rgb.backward() - colour / density / main grid / pose - is queued "
                      "before the host waits for the cost matrices, the panoptic half follows the assignment", "voxel", 2, set(all_ch), 2, 20))
            specs = specs + (("post_prune_all_assign_host_solver", late + "; ONE loss.backward(), LinAssignmentThingsLoss(solver='scipy'): cost matrices copied to the host, the step waits "
                              "for them and for SciPy (the default of round 5; the other two lines solve the assignment on the device)", "voxel", 2, set(all_ch), 4, 20),)
            specs = specs + (("post_prune_all_assign_one_backward_trained_head", late + "; ONE loss.backward(), after 300 training steps of this very objective: the instance head then tells the "
                              "labels apart, each label prefers its own columns and the device assignment takes a few dozen search passes per image instead of the ~300 of an UNTRAINED head "
                              "(every label's mean row about the same: each new label's search walks through all the earlier ones) - what the late epochs of a run look like", "voxel", 2, set(all_ch), 5, 20),)
            specs = specs + (("dense_rgb_trained_scene", "the dense regime on a LEARNABLE scene (analytic textured sphere, white background) after 300 training steps: most "
                              "samples in front of the surface are empty space with sigma = relu(pre) = 0 exactly, their gradients are exactly zero and the encoders' backward "
                              "skips their waves (bin pass) and row requests (position gradient), bit-identically; the untrained random scene of the other lines has no such samples", "ray", 512, {"rgb", "depth"}, 3, 6),)
            for tag, what, rm, smp, chans, assign, n_steps in specs:
                j = Job(rays_n=total, samples=smp, grid="permuto", channels=chans, raymarch=rm, pose=True)
                trained = assign == 3
                if trained:
                    assign = 0
                    with torch.no_grad():
                        r0 = j.rays()
                    rgb_gt, hit = sphere_gt(r0.origins, r0.dirs)
                    j.gt["rgb"] = rgb_gt.to(dev)
                    for _ in range(300):
                        j.step()
                host_solver, trained_head = assign == 4, assign == 5
                if host_solver or trained_head:
                    assign = 1
                if assign:
                    j.lin_assign, j.images, j.points_fn, j.seg_reg = LinAssignmentThingsLoss(outlier_rejection=True, solver="scipy" if host_solver else None), images, j.pose.points_3d, True
                if assign == 2:
                    j.overlap, j.tracer.graph_split = True, True
                for _ in range(304 if trained_head else 4):         # step 0 learns the sample count, step 1 captures, 2 - 3 replay
                    j.step()
                d, _ = j.timed(n_steps)
                ms = d / n_steps * 1e3
                m = j.samples_per_step()
                ent = dict(workload=what, ms_per_step=round(ms, 3), rays_s=round(total / ms * 1e3, 1), samples_per_step=int(m), steps=n_steps,
                           hip_graphs=j.graph_stats())
                if rm == "voxel":
                    ent["occupied_fraction"] = round(j.occupied, 4)
                if trained:
                    with torch.no_grad():
                        rr = j.rays()
                        mo = j.nef.grid.raymarch(rr, level=None, num_samples=smp, raymarch_type=rm)
                        dens = j.nef(coords=mo[2], ray_d=rr.dirs.index_select(0, mo[0]), channels="density").reshape(-1)
                    ent["pretraining_steps"] = 300
                    ent["samples_with_sigma_exactly_zero"] = round(float((dens == 0).float().mean()), 4)
                    ent["rays_hitting_the_sphere"] = round(float(hit.float().mean()), 4)
                    del dens, mo
                # eager pass with events around every C-ABI call: where the step's device time goes, with bytes and fractions of the roofs
                was, j.tracer.use_graphs = j.tracer.use_graphs, False
                for _ in range(2):
                    j.step()
                n_bd = 3
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                ops.profile_start()
                for _ in range(n_bd):
                    j.step()
                prof = ops.profile_stop()
                ent["eager_ms_per_step_with_events"] = round((time.perf_counter() - t0) / n_bd * 1e3, 3)
                j.tracer.use_graphs = was
                model = algorithmic_model("permuto", m, total, chans, 24, 2, 4, args.precision == "bf16")
                table, _, _ = kernel_table(prof, n_bd, model)
                ent["entry_points"] = {k: v for k, v in table.items() if v["ms_per_step"] >= 0.003}
                ent["device_ms_per_step"] = round(float(sum(np.sum(v) for v in prof.values())) / n_bd, 3)
                if assign:
                    ent["labels_per_image"] = int((torch.unique(j.gt["inst_ids"].reshape(images, -1)[0]) > 0).sum())
                regimes[tag] = ent
                j.close()
                del j
                torch.cuda.empty_cache()
            out["regimes"] = regimes
            return out

        if world == 1 and default_cfg:
            # ---- every other single-GPU BASELINE configuration, each a short run (configs[1] is the headline above)
            cfgs = []
            cfgs.append(short_run("configs[0] on the GPU: hash L=16 T=2^19, 256 rays x 64 samples, rgb", 20, 5, key="cfg0_hash_256x64_rgb",
                                  rays_n=256, samples=64, grid="hash", channels={"rgb"}))
            cfgs.append(short_run("configs[2]: hash L=16 T=2^19 (16..2048) + fused MFMA decoders, 4096 rays x 512, all channels", 20, 5, key="cfg2_hash_4096x512_all",
                                  rays_n=4096, samples=512, grid="hash", channels=all_ch))
            cfgs.append(short_run("other shipped head shapes (sem_num_layers 2 / inst_num_layers 1: lin_assign_delta_app.yaml:117,120 and three more YAMLs): 4096 rays x 512, "
                                  "permuto, all channels - the two-layer 200-way head takes the generic decoder kernels", 10, 3, key="heads_2_1_permuto_4096x512_all",
                                  rays_n=4096, samples=512, grid="permuto", channels=all_ch, heads=(2, 1)))
            cfgs.append(short_run("configs[3] on ONE GPU: 6 images x 4096 rays, ba_pipeline pose-opt, permuto, all channels", 5, 2, key="cfg3_one_gpu_24576x512_pose_all",
                                  rays_n=24576, samples=512, grid="permuto", channels=all_ch, pose=True))
            cfgs.append(short_run("configs[4] per-GPU shard: 131072 rays x 64 samples, permuto, fp16 tables + bf16 features, rgb", 10, 3, key="cfg4_shard_permuto_fp16",
                                  rays_n=131072, samples=64, grid="permuto", channels={"rgb"}, table_dtype="fp16"))
            cfgs.append(short_run("configs[4] per-GPU shard: 131072 rays x 64 samples, permuto, fp32 tables, rgb", 10, 3, key="cfg4_shard_permuto_fp32",
                                  rays_n=131072, samples=64, grid="permuto", channels={"rgb"}))
            cfgs.append(short_run("configs[4] per-GPU shard: 131072 rays x 64 samples, hash L=16 T=2^19 res 16..1024, fp16 tables, rgb", 10, 3, key="cfg4_shard_hash_fp16",
                                  rays_n=131072, samples=64, grid="hash", channels={"rgb"}, finest=1024, table_dtype="fp16"))
            cfgs.append(short_run("post-prune regime (f3): voxel march, %.0f %% occupancy, 2 samples per voxel, permuto, all channels"
                                  % (100 * args.occupancy), 20, 5, key="post_prune_4096_all", rays_n=4096, samples=2, grid="permuto", channels=all_ch, raymarch="voxel"))
            cfgs.append(short_run("post-prune regime, rgb only (epochs 201 - 600 of best.yaml: voxel march from 201, panoptic heads from 601): %.0f %% occupancy, permuto"
                                  % (100 * args.occupancy), 20, 5, key="post_prune_4096_rgb", rays_n=4096, samples=2, grid="permuto", channels={"rgb"}, raymarch="voxel"))
            line["configs"] = cfgs
            # ---- the 4096-ray, pose-free epoch-weighted number of rounds 3 - 4 (kept for continuity; NOT a step best.yaml executes - its
            #      batch is 6 images and its extrinsics are trainable in every epoch - see best_yaml_step / schedule_weighted below)
            try:
                vox_all = next(c for c in cfgs if c["name"].startswith("post-prune regime (f3)"))
                vox_rgb = next(c for c in cfgs if c["name"].startswith("post-prune regime, rgb only"))
                parts = [(200, line["rgb_only"]["ms_per_step"]), (400, vox_rgb["ms_per_step"]), (200, vox_all["ms_per_step"])]
                ms = sum(w * t for w, t in parts) / sum(w for w, _ in parts)
                line["schedule_weighted_4096_no_pose"] = dict(
                    ms_per_step=round(ms, 3), rays_s=round(args.rays / ms * 1e3, 1), epochs_and_ms=[dict(epochs=w, ms_per_step=t) for w, t in parts],
                    note="4096 rays per step, NO pose optimisation, fixed instance targets: 200 epochs dense rgb-only, 400 post-prune rgb-only, 200 "
                         "post-prune all channels; synthetic occupancy %.0f %% after the prune.  best.yaml itself runs 6 x 4096 rays with trainable "
                         "extrinsics in every epoch: see schedule_weighted" % (100 * args.occupancy))
            except (StopIteration, KeyError):
                pass
            # ---- the step `train.sh` -> configs/bup20/best.yaml actually runs, in its three regimes: batch_size 6 (:156) x 4096 rays (:19) = 24 576
            #      rays per step; optimize_extrinsics (:184) with extrinsics_epoch_start 0 / _end 900 > epochs 800 (:158-160): pc_nerf/trainer.py:308
            #      keeps the extrinsics trainable in EVERY epoch, so every step goes through pc_nerf/ba_pipeline.py:85-92 with rays that carry a
            #      gradient; inst_outlier_rejection (:106) adds 'depth' to the channels of every step (trainer.py:432); 'ray' march with 512 steps
            #      until voxel_raymarch_epoch_start = prune_every = 201 (:34,:187), voxel march with 2 samples per voxel after it; panoptic heads
            #      from sem_epoch_start = inst_epoch_start = 601 (:89,:165), with the instance term of trainer.py:483-533 = LinAssignmentThingsLoss
            #      (outlier rejection from the depth's 3-D points, :508-518) + segment_consistency_regularizer (:525-527, weight 1.0 - trainer.py:93).
            line["best_yaml_step"] = best_yaml_regimes()
            try:
                reg = line["best_yaml_step"]["regimes"]
                parts = [(200, reg["dense_rgb"]["ms_per_step"]), (400, reg["post_prune_rgb"]["ms_per_step"]),
                         (200, min(reg["post_prune_all_assign"]["ms_per_step"], reg["post_prune_all_assign_one_backward"]["ms_per_step"]))]
                ms = sum(w * t for w, t in parts) / sum(w for w, _ in parts)
                line["schedule_weighted"] = dict(
                    ms_per_step=round(ms, 3), rays_per_step=6 * 4096, rays_s=round(6 * 4096 / ms * 1e3, 1),
                    epochs_and_ms=[dict(epochs=w, ms_per_step=t) for w, t in parts],
                    note="epoch-weighted mean step of a configs/bup20/best.yaml run, every step 6 images x 4096 rays with pose optimisation (BAPipeline, "
                         "trainable extrinsics) and channels rgb + depth: 200 epochs dense 'ray' march, 400 post-prune voxel march, 200 post-prune with the "
                         "panoptic heads + LinAssignmentThingsLoss(outlier_rejection=True) on the 6-image batch + segment_consistency_regularizer; "
                         "synthetic occupancy %.0f %% after the prune; HIP graphs %s" % (100 * args.occupancy, args.graphs))
            except KeyError:
                pass
        if world == 1 and default_cfg:
            line["render"] = render_image_line(args, dev, all_ch, out_bytes)
            line["render_pruned"] = render_image_line(args, dev, all_ch, out_bytes, pruned=True)
        if world > 1 and default_cfg:
            # ---- weak scaling of the regimes that dominate a best.yaml schedule (4096 rays per GPU): the table exchange costs the same whatever the batch,
            #      so the short post-prune steps are where it shows; grad_sync = what GradSync(comm_dtype="auto") decided for each
            line["weak_regimes"] = [
                short_run("weak: post-prune voxel march, rgb only, 4096 rays per GPU", 30, shard.AUTO_WARM + 6, rays_n=4096, samples=2, grid="permuto",
                          channels={"rgb"}, raymarch="voxel"),
                short_run("weak: post-prune voxel march, all channels, 4096 rays per GPU", 30, shard.AUTO_WARM + 6, rays_n=4096, samples=2, grid="permuto",
                          channels=all_ch, raymarch="voxel"),
                # the same two regimes with the touched-rows exchange (unmeasured on hardware so far: the pair of lines is the A/B)
                short_run("weak: post-prune voxel march, rgb only, 4096 rays per GPU, sparse table exchange", 30, shard.AUTO_WARM + 10, rays_n=4096, samples=2,
                          grid="permuto", channels={"rgb"}, raymarch="voxel", sparse="bounded"),
                short_run("weak: post-prune voxel march, all channels, 4096 rays per GPU, sparse table exchange", 30, shard.AUTO_WARM + 10, rays_n=4096, samples=2,
                          grid="permuto", channels=all_ch, raymarch="voxel", sparse="bounded")]
            # ---- strong scaling, BASELINE configs[3]: one 24 576-ray step (6 images, pose-opt) split over the ranks
            total = 6 * 4096
            ent = short_run("configs[3]: 6 images x 4096 rays, ba_pipeline pose-opt, %d rays per GPU" % (total // world), 10, 3,
                            rays_n=total // world, samples=512, grid="permuto", channels=all_ch, pose=True, total_rays=total)
            ent["scaling"] = "strong"
            line["strong"] = ent
            # ---- validation render sharded over the ranks + ONE all_gather of the buffers (shard.render_sharded)
            import pagnerf_amd
            nef = make_model(args, dev, seed=0)
            pipe = pagnerf_amd.Pipeline(nef, make_tracer(args, "ray", 512))
            n_val = 8192 * world
            rays_all, _ = make_rays(n_val, dev, seed=5)
            with torch.no_grad():
                for _ in range(2):
                    shard.render_sharded(pipe, rays_all, channels=sorted(all_ch))
                barrier()
                t0 = time.perf_counter()
                for _ in range(5):
                    rb = shard.render_sharded(pipe, rays_all, channels=sorted(all_ch))
                barrier()
                d = max_over_ranks(time.perf_counter() - t0) / 5
                lo, hi = shard.shard_bounds(n_val, rank, world)
                torch.manual_seed(4242 + rank)          # the march draws its jitter from the device generator: same seed, same samples
                rb = shard.render_sharded(pipe, rays_all, channels=sorted(all_ch))
                torch.manual_seed(4242 + rank)
                local = pipe(rays=rays_all[lo:hi], channels=sorted(all_ch))
                same = bool(torch.equal(rb.rgb[lo:hi], local.rgb))
            line["render_sharded"] = dict(rays=n_val, ms=round(d * 1e3, 3), rays_s=round(n_val / d, 1), gathered_bytes_per_rank=n_val * 211 * 4,
                                          gathered_equals_local=same)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(64, 512)
    elif rank == 0:
        line["cpu_baseline"] = None
    if rank == 0:
        emit(line)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    argv = sys.argv[1:]
    args = parse(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return spawn_ranks(args, argv)
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
