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

The JSON line also carries
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


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--channels", default="all", choices=["all", "rgb"])
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--samples", type=int, default=512)
    ap.add_argument("--grid", default="permuto", choices=["permuto", "hash"])
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--raymarch", default="ray", choices=["ray", "voxel"])
    ap.add_argument("--occupancy", type=float, default=0.1, help="voxel mode: fraction of occupied 128^3 cells after the synthetic prune")
    ap.add_argument("--pose-opt", action="store_true", help="BAPipeline: rays from learnable camera extrinsics (configs[3])")
    ap.add_argument("--images", type=int, default=6, help="--pose-opt: images per step (rays are split evenly over them)")
    ap.add_argument("--fp32-coords", action="store_true", help="permuto grids: skip the fp16 coordinate rounding of the reference's autocast")
    ap.add_argument("--sustain-steps", type=int, default=300, help="extra timed region after the K steps (0 = off)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-aux", action="store_true", help="skip the rgb-only / sustained / configs / strong measurements")
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
def make_model(args, dev, seed, grid=None, num_lods=None, log2T=None, finest=None):
    import torch
    import pagnerf_amd
    grid = grid or args.grid
    torch.manual_seed(seed)
    common = dict(feature_dim=2, num_classes=6, num_instances=200, sem_num_layers=1, sem_softmax=True, inst_num_layers=2,
                  inst_softmax=True, panoptic_features_type="delta", hidden_dim=64, num_layers=1, view_multires=4,
                  precision=args.precision, blas_level=7)
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
            g.tables.data.normal_(0, 1e-2)
    return nef.to(dev)


def make_tracer(args, raymarch=None, samples=None):
    import pagnerf_amd
    rm = raymarch or args.raymarch
    if rm == "voxel":            # after trainer.py:362-366: 2 samples per intersected occupied voxel (best.yaml:31), ray_max_travel 6 x scale
        return pagnerf_amd.PanopticPackedRFTracer(raymarch_type="voxel", num_steps=2, bg_color="white", ray_max_travel=6.0)
    return pagnerf_amd.PanopticPackedRFTracer(raymarch_type="ray", num_steps=samples or args.samples, bg_color="white")


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
    return pagnerf_amd.Rays(o.to(dev), d.to(dev), dist_min=0.0, dist_max=1.9), {k: v.to(dev) for k, v in gt.items()}


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
        self.o, self.d, self.cam = o[lo:hi].to(dev), d[lo:hi].to(dev), cam[lo:hi].clamp(max=C - 1).to(dev)

    def rays(self):
        return self.pipe.transform_rays_indexed(self.o, self.d, self.cam)

    def parameters(self):
        return [self.pipe.camera_extrinsics]


def make_optimizer(nef, extra=()):
    import torch
    grid_params = [p for n, p in nef.named_parameters() if "grid" in n]
    rest = [p for n, p in nef.named_parameters() if "grid" not in n]
    groups = [dict(params=grid_params, lr=1e-3 * 100), dict(params=rest, lr=1e-3)]     # best.yaml:103,108 ; trainer.py:272-281
    if extra:
        groups.append(dict(params=list(extra), lr=1e-4))
    try:
        return torch.optim.Adam(groups, eps=1e-15, fused=True)                         # config_parser.py:672
    except Exception:
        return torch.optim.Adam(groups, eps=1e-15)


def train_step(nef, tracer, opt, rays, gt, channels, world, sync=None):
    import torch
    import torch.nn.functional as F
    opt.zero_grad(set_to_none=True)
    if callable(rays):
        rays = rays()            # pose optimisation: this step's rays from the current extrinsics
    rb = tracer(nef, channels=channels, rays=rays, stage="train")
    # trainer.py:443-446 / best.yaml:116 rgb L1; trainer.py:465-467 nll_loss(log(p + 1e-27), gt); the instance term stands for the
    # per-image linear-assignment NLL (trainer.py:499-520 -> loss/lin_assignment_things.py:80), same arithmetic on a fixed target.
    # pagnerf_amd.loss.render_loss evaluates exactly that sum in one launch (and its gradients in one more).
    from pagnerf_amd.loss import render_loss, NllTerm
    if os.environ.get("PAG_BENCH_TORCH_LOSS"):
        loss = 10.0 * torch.abs(rb.rgb - gt["rgb"]).mean()
        if "semantics" in channels:
            loss = loss + 0.1 * F.nll_loss(torch.log(rb.semantics + 1e-27), gt["sem"], reduction="mean")
            loss = loss + 1000.0 * F.nll_loss(torch.log(rb.inst_embedding + 1e-27), gt["inst"], reduction="mean")
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


def cpu_baseline(n_rays, n_samples, budget_s=20.0):
    """The oracle's torch-CPU restatement of the reference's hash_grid_torch path, forward + backward."""
    import numpy as np
    import torch
    from oracle import hash_encode as oh, decoders as od, render as orr
    # torch-CPU ops on these small tensors scale badly past a few dozen threads (256 threads on the GPU
    # box's host ran 400x slower than 8): use at most 32 and report the count actually used.
    torch.set_num_threads(min(32, os.cpu_count() or 1))
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
    o = torch.cat([(torch.rand(n_rays, 2) - 0.5) * 0.6, torch.full((n_rays, 1), 0.95)], 1)
    dr = torch.nn.functional.normalize(torch.cat([(torch.rand(n_rays, 2) - 0.5) * 0.7, -torch.ones(n_rays, 1)], 1), dim=-1)
    gt = torch.rand(n_rays, 3)

    def step():
        ridx, pidx, samples, depths, deltas, boundary = orr.raymarch_ray(o, dr, 0.0, 1.9, n_samples, torch.rand(n_rays, n_samples))
        feats, _ = oh.hash_encode(samples[:, 0], tables, res, log2T)
        out = od.nef_forward(feats, None, dr[ridx], params, {"rgb"})
        comp = orr.composite(n_rays, ridx, boundary, out["density"], deltas, rgb=out["rgb"])
        loss = 10.0 * torch.abs(comp["rgb"] - gt).mean()
        loss.backward()
    t_w = time.perf_counter()
    step()                                   # warm-up (also sizes the budget)
    t_w = time.perf_counter() - t_w
    note = ("DIFFERENT WORKLOAD from the GPU line: %d rays x %d samples, HASH grid L=16 T=2^19 (the reference's CPU-runnable "
            "grids/hash_grid_torch.py, restated op for op in oracle/), density+colour decoders only (no panoptic heads, no delta grid), "
            "compositing, rgb L1, forward+backward, torch-CPU" % (n_rays, n_samples))
    if t_w > budget_s:                       # pathological host: report the single step rather than overrun
        return dict(value=n_rays / t_w, unit="rays/s", cores=torch.get_num_threads(), kind="port",
                    sample=note + "; 1 step of %.1f s (warm-up only; host too slow for more)" % t_w)
    t0, n = time.perf_counter(), 0
    while True:
        step()
        n += 1
        if time.perf_counter() - t0 > budget_s or n >= 50:
            break
    dt = (time.perf_counter() - t0) / n
    return dict(value=n_rays / dt, unit="rays/s", cores=torch.get_num_threads(), kind="port",
                sample=note + "; %d timed steps of %.2f s" % (n, dt))


# -------------------------------------------------------------------------- algorithmic bytes / flops per entry point
def algorithmic_model(grid, M, N, channels, L_, F_, verts, bf16):
    """Per C-ABI entry point and STEP: (algorithmic HBM bytes, useful MFMA flops).  Bytes = every tensor the launch must read or write
    once at the dtypes of the production path (SURVEY 8d per-sample figures x M); tables count as gathered bytes (rows x F x 4)."""
    s = 2 if bf16 else 4
    C = L_ * F_
    gather = L_ * verts * F_ * 4
    pan = "semantics" in channels
    enc_bwd = (12 + C * s + 2 * gather) * M * (2 if pan else 1)                                  # grad row + xyz + RMW of the gathered rows
    # decoders (34 560 MAC per sample with C=6, I=200 - SURVEY a8): density 48->64->16, colour 43->64->64->3, sem 48->64->6, inst 48->64->64->200
    mac = dict(density=C * 64 + 64 * 16, colour=43 * 64 + 64 * 64 + 64 * 3, sem=C * 64 + 64 * 6, inst=C * 64 + 64 * 64 + 64 * 200)
    used = ["density", "colour"] + (["sem", "inst"] if pan else [])
    flops_fwd = 2 * M * sum(mac[k] for k in used)
    # forward traffic: inputs + outputs + the hidden activations saved for the backward ([M,64] per hidden layer)
    io = dict(density=(C + 16 + 64) * s, colour=(16 + 64 + 64) * s + 3 * 4 + 4, sem=(C + 64) * s + 8, inst=(C + 64 + 64) * s + 8)
    mlp_fwd = M * sum(io[k] for k in used)
    # backward-data: reads saved activations + upstream gradient, writes dz per layer (+ dx for density / heads)
    iob = dict(density=(64 + 16 + 64 + 16 + C) * s, colour=(64 + 64 + 3 + 64 + 64 + 3 + 16) * s, sem=(64 + 64 + 6 + C) * s + 8,
               inst=(64 + 64 + 64 + 64 + 200 + C) * s + 8)
    mlp_bwd = M * sum(iob[k] for k in used)
    # weight gradients: every layer streams its dz and its input once
    iow = dict(density=(C + 64 + 64 + 16) * s, colour=(16 + 64 + 64 + 64 + 64 + 3) * s, sem=(C + 64 + 64 + 6) * s,
               inst=(C + 64 + 64 + 64 + 64 + 200) * s)
    wgrad = M * sum(iow[k] for k in used)
    comp = M * (4 + 4 + 4 + 12 + 4) + N * 24
    return {
        "pag_%s_encode_fwd" % grid: ((12 + gather + C * s) * M, 0),
        "pag_%s_encode_fwd_add" % grid: ((12 + gather + 2 * C * s) * M, 0),
        "pag_%s_encode_bwd_set" % grid: (enc_bwd, 0),
        "pag_mlp_fwd": (mlp_fwd, flops_fwd),
        "pag_mlp_bwd": (mlp_bwd, flops_fwd),            # dX = dZ W: the same MACs as the forward (first-layer dx of the colour decoder's PE excluded)
        "pag_mlp_wgrad_batch": (wgrad, flops_fwd),      # dW = dZ^T A: the same MACs again
        "pag_composite_fwd": (comp, 0),
        "pag_composite_bwd": (comp + M * 16, 0),
        "pag_head_composite_fwd": (M * (64 * s + 8 + 4) * 1 + N * 206 * 4, 2 * M * 64 * 200),
    }


# -------------------------------------------------------------------------------------------------- dry run (CPU, gloo)
def dry_run_rank(args, world, rank):
    """No kernels: the process group, shard.GradSync / all_gather_render on CPU tensors, the timing protocol and the JSON line."""
    import torch
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
    sync = shard.GradSync(params, early=[params[0]]) if world > 1 else None
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
        assert torch.allclose(params[0].grad, torch.full((64, 8), mean)), "GradSync over the bench's process group gave a wrong mean"
        lo, hi = shard.shard_bounds(n_local * world, rank, world)
        rb = shard.all_gather_render(RenderBuffer(rgb=torch.arange(lo, hi, dtype=torch.float32)[:, None].repeat(1, 3)), n_local * world)
        assert torch.equal(rb.rgb[:, 0], torch.arange(n_local * world, dtype=torch.float32))
    if rank == 0:
        print(json.dumps(dict(metric="rays/sec (train step) on BUP20-shape scene", value=0.0, unit="rays/s", n_gpus=world,
                              steps=args.steps, warmup=args.warmup, ms_per_step=round(dt / max(args.steps, 1) * 1e3, 3),
                              higher_is_better=True, scaling="weak", vs_baseline=None, dtype=args.precision, data="synthetic",
                              config=dict(workload="DRY RUN: no kernels, gloo on CPU tensors - plumbing check only"),
                              dry_run=True, rccl_ranks_seen=int(seen.item()), backend="gloo")), flush=True)
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

        def __init__(self, rays_n, samples, grid, channels, raymarch="ray", pose=False, total_rays=None, seed=0, **mk):
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
            self.sync = None
            if world > 1:
                early = [self.nef.delta_grid.tables] if hasattr(self.nef, "delta_grid") else []
                self.sync = shard.GradSync(list(self.nef.parameters()) + list(extra), early=early)

        def step(self, channels=None):
            return train_step(self.nef, self.tracer, self.opt, self.rays, self.gt, channels or self.channels, world, self.sync)

        def timed(self, n_steps, channels=None, profile=None):
            barrier()
            if profile is not None:
                ops.profile_start(only=profile)
            t0 = time.perf_counter()
            for _ in range(n_steps):
                self.step(channels)
            barrier()
            dt = time.perf_counter() - t0
            prof = ops.profile_stop() if profile is not None else None
            return max_over_ranks(dt), prof

        def samples_per_step(self):
            with torch.no_grad():
                r = self.rays() if callable(self.rays) else self.rays
                out = self.nef.grid.raymarch(r, level=None, num_samples=self.tracer.num_steps, raymarch_type=self.tracer.raymarch_type)
            return int(out[2].shape[0] * (out[2].shape[1] if out[2].dim() == 3 else 1))

        def close(self):
            if self.sync is not None:
                self.sync.remove()

    all_ch = {"rgb", "depth", "semantics", "inst_embedding"}
    channels = all_ch if args.channels == "all" else {"rgb"}
    job = Job(args.rays, args.samples, args.grid, channels, raymarch=args.raymarch, pose=args.pose_opt)
    enc_name = "pag_%s_encode_fwd" % args.grid

    for _ in range(args.warmup):
        job.step()
    # HIP events around the roofline kernel only; the full per-entry-point breakdown comes from a separate pass
    dt, prof = job.timed(args.steps, profile=None if os.environ.get("PAG_BENCH_PROFILE_ALL") else {enc_name})

    M = job.samples_per_step() if (args.raymarch == "voxel" or args.pose_opt) else args.rays * args.samples
    L_, F_ = (24, 2) if args.grid == "permuto" else (16, 2)
    verts = 4 if args.grid == "permuto" else 8
    out_bytes = 2 if args.precision == "bf16" else 4
    bytes_per_sample = 12 + L_ * verts * F_ * 4 + L_ * F_ * out_bytes       # xyz + table gathers + feature row
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
                        algorithmic_bytes_per_launch=bytes_per_sample * M)

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
                    raymarch=args.raymarch, half_coords=(args.grid == "permuto" and not args.fp32_coords),
                    parallelism="ray-sharded data parallel x%d" % world),
        rccl_ranks_seen=ranks_seen, backend=backend, roofline=roofline)

    if not args.no_aux:
        # ---- per-entry-point device time: a separate, untimed pass with events around every C-ABI call
        n_bd = max(1, min(5, args.steps))
        barrier()
        ops.profile_start()
        for _ in range(n_bd):
            job.step()
        prof_all = ops.profile_stop()
        model = algorithmic_model(args.grid, M, args.rays, channels, L_, F_, verts, args.precision == "bf16")
        kernels, mfma_ms, mfma_flops = {}, 0.0, 0.0
        for k, v in sorted(prof_all.items()):
            ms = float(np.sum(v)) / n_bd
            ent = dict(calls_per_step=len(v) / n_bd, ms_per_step=round(ms, 4))
            if k in model and ms > 0:
                by, fl = model[k]
                ent.update(algorithmic_bytes=int(by), hbm_frac=round(by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
                if fl:
                    ent.update(mfma_tflops=round(fl / (ms * 1e-3) / 1e12, 2), mfma_frac=round(fl / (ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 5))
                    mfma_ms += ms
                    mfma_flops += fl
            kernels[k.replace("pag_", "")] = ent
        line["kernels"] = kernels
        if mfma_ms:
            line["mfma_util"] = dict(
                algorithmic_tflops=round(mfma_flops / (mfma_ms * 1e-3) / 1e12, 2), peak_tflops=MFMA_PEAK_TFLOPS,
                frac=round(mfma_flops / (mfma_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 5), decoder_ms_per_step=round(mfma_ms, 4),
                note="useful decoder MACs (34 560 / sample x fwd, bwd-data, wgrad) over the decoder entry points' device time; K <= 64 "
                     "MLPs are activation-traffic bound (see kernels.*.hbm_frac); PMC MFMA-busy cycles: profiles/README.md")
        # ---- sustained: >= 2 s of the same step; the clocks / temperature of a 0.14 s burst are not what training sees
        if args.sustain_steps > 0:
            half = args.sustain_steps // 2
            d1, _ = job.timed(args.sustain_steps - half)
            d2, _ = job.timed(half)
            line["sustained"] = dict(steps=args.sustain_steps, seconds=round(d1 + d2, 3),
                                     ms_per_step=round((d1 + d2) / args.sustain_steps * 1e3, 3),
                                     ms_per_step_last_half=round(d2 / max(half, 1) * 1e3, 3),
                                     value_last_half=round(world * args.rays * half / d2, 1), unit="rays/s")
        # ---- rgb-only regime (epochs < 601, best.yaml:89)
        if args.channels == "all":
            for _ in range(2):
                job.step({"rgb"})
            n_aux = max(3, args.steps // 2)
            dt_rgb, _ = job.timed(n_aux, {"rgb"})
            line["rgb_only"] = dict(workload="same scene, channels {rgb} only (epochs < 601, best.yaml:89)",
                                    value=round(world * args.rays * n_aux / dt_rgb, 1), unit="rays/s", ms_per_step=round(dt_rgb / n_aux * 1e3, 3))
        default_cfg = (args.grid, args.rays, args.samples, args.raymarch, args.pose_opt, args.channels) == ("permuto", 4096, 512, "ray", False, "all")
        job.close()
        del job
        torch.cuda.empty_cache()

        def short_run(name, n_steps, warm, **kw):
            j = Job(**kw)
            for _ in range(warm):
                j.step()
            d, p = j.timed(n_steps, profile={"pag_%s_encode_fwd" % kw["grid"]})
            m = j.samples_per_step() if (kw.get("raymarch") == "voxel" or kw.get("pose")) else kw["rays_n"] * kw["samples"]
            rays_total = kw.get("total_rays") or kw["rays_n"] * world
            ms = d / n_steps * 1e3
            e = p.get("pag_%s_encode_fwd" % kw["grid"], [])
            lv, vt = (24, 4) if kw["grid"] == "permuto" else (kw.get("num_lods") or 16, 8)
            bps = 12 + lv * vt * 2 * 4 + lv * 2 * out_bytes
            ent = dict(name=name, ms_per_step=round(ms, 3), rays_s=round(rays_total / ms * 1e3, 1), samples_per_step=int(m), steps=n_steps,
                       encode_frac=round(bps * m / (float(np.mean(e)) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if e else None)
            if kw.get("raymarch") == "voxel":
                ent["occupied_fraction"] = round(j.occupied, 4)
            j.close()
            del j
            torch.cuda.empty_cache()
            return ent

        if world == 1 and default_cfg:
            # ---- every other single-GPU BASELINE configuration, each a short run (configs[1] is the headline above)
            cfgs = []
            cfgs.append(short_run("configs[0] on the GPU: hash L=16 T=2^19, 256 rays x 64 samples, rgb", 20, 5,
                                  rays_n=256, samples=64, grid="hash", channels={"rgb"}))
            cfgs.append(short_run("configs[2]: hash L=16 T=2^19 (16..2048) + fused MFMA decoders, 4096 rays x 512, all channels", 20, 5,
                                  rays_n=4096, samples=512, grid="hash", channels=all_ch))
            cfgs.append(short_run("configs[3] on ONE GPU: 6 images x 4096 rays, ba_pipeline pose-opt, permuto, all channels", 5, 2,
                                  rays_n=24576, samples=512, grid="permuto", channels=all_ch, pose=True))
            cfgs.append(short_run("configs[4] per-GPU shard: 131072 rays x 64 samples, permuto, rgb", 10, 3,
                                  rays_n=131072, samples=64, grid="permuto", channels={"rgb"}))
            cfgs.append(short_run("configs[4] per-GPU shard: 131072 rays x 64 samples, hash L=16 T=2^19 res 16..1024, rgb", 10, 3,
                                  rays_n=131072, samples=64, grid="hash", channels={"rgb"}, finest=1024))
            cfgs.append(short_run("post-prune regime (f3): voxel march, %.0f %% occupancy, 2 samples per voxel, permuto, all channels"
                                  % (100 * args.occupancy), 20, 5, rays_n=4096, samples=2, grid="permuto", channels=all_ch, raymarch="voxel"))
            line["configs"] = cfgs
        if world > 1 and default_cfg:
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
                local = pipe(rays=rays_all[lo:hi], channels=sorted(all_ch))
                same = bool(torch.equal(rb.rgb[lo:hi], local.rgb))
            line["render_sharded"] = dict(rays=n_val, ms=round(d * 1e3, 3), rays_s=round(n_val / d, 1), gathered_bytes_per_rank=n_val * 211 * 4,
                                          gathered_equals_local=same)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(64, 512)
    elif rank == 0:
        line["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(line), flush=True)
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
