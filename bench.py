#!/usr/bin/env python3
"""rays/s of one PAg-NeRF train step through the HIP hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--channels all|rgb] [--rays 4096] [--samples 512]

Workload at N = 1 (BASELINE.json configs[1]): PanopticDeltaNeF + permutohedral grids (L = 24, F = 2,
T = 2^18, main + delta), 4096 rays x 512 'ray'-mode samples (M = 2 097 152 packed samples, dense
occupancy), bf16 MFMA decoders on bf16 features with fp32 tables / accumulation / compositing.
A step = ray march -> encode -> decoders -> compositing -> loss (rgb L1 x10, + semantic / instance
NLL against fixed synthetic labels when the panoptic heads are on; pc_nerf/trainer.py:443-480) ->
backward through every kernel -> Adam (eps 1e-15, grid lr x100) - nothing is cached between steps.
For N > 1 (launched by torch.distributed.run, one rank per GPU) every rank marches its own 4096-ray
shard against replicated parameters and the gradients are summed with one flat RCCL all-reduce:
weak scaling, value = all ranks' rays / max-over-ranks time.

The JSON line also carries
  roofline      the permutohedral encode forward launch (the grid-interpolate kernel north_star sets
                the 40 % HBM target on): algorithmic bytes per launch / its mean duration measured
                with HIP events on the launch stream inside the timed region
  cpu_baseline  the CPU oracle's restatement of the reference's grids/hash_grid_torch.py path
                (encode -> decoders -> compositing, forward + backward) on this host's cores,
                on a bounded sample of the same workload (kind "port").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.nn.functional as F
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--channels", default="all", choices=["all", "rgb"])
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--samples", type=int, default=512)
    ap.add_argument("--grid", default="permuto", choices=["permuto", "hash"])
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-aux", action="store_true", help="skip the auxiliary rgb-only measurement")
    return ap.parse_args()


def make_model(args, dev, seed):
    import pagnerf_amd
    torch.manual_seed(seed)
    common = dict(feature_dim=2, num_classes=6, num_instances=200, sem_num_layers=1, sem_softmax=True, inst_num_layers=2,
                  inst_softmax=True, panoptic_features_type="delta", hidden_dim=64, num_layers=1, view_multires=4,
                  precision=args.precision, blas_level=7)
    if args.grid == "permuto":   # configs/bup20/best.yaml:47-65
        nef = pagnerf_amd.PanopticDeltaNeF(grid_type="PermutoGrid", num_lods=24, capacity_log_2=18, delta_capacity_log_2=18,
                                           coarsest_scale=1.0, finest_scale=1e-4, **common)
        for g in (nef.grid, nef.delta_grid):
            g.init_from_scales(tables=torch.randn(24, 2 ** 18, 2) * 1e-2)
    else:                        # BASELINE.json configs[2]: 16-level hash grid, T = 2^19
        nef = pagnerf_amd.PanopticDeltaNeF(grid_type="HashGridTorch", num_lods=16, codebook_bitwidth=19, **common)
        for g in (nef.grid, nef.delta_grid):
            g.init_from_resolutions([16] * 15 + [2048])
            g.tables.data.normal_(0, 1e-2)
    nef = nef.to(dev)
    tracer = pagnerf_amd.PanopticPackedRFTracer(raymarch_type="ray", num_steps=args.samples, bg_color="white")
    return nef, tracer


def make_rays(n, dev, seed):
    """BUP20-shaped synthetic view: downward-looking pinhole rays that stay inside [-1,1]^3
    (near 0, datasets/formats/bup20.py:249; far scaled so every sample survives the dense BLAS)."""
    import pagnerf_amd
    g = torch.Generator().manual_seed(seed)
    o = torch.cat([(torch.rand(n, 2, generator=g) - 0.5) * 0.6, torch.full((n, 1), 0.95)], 1)
    d = torch.cat([(torch.rand(n, 2, generator=g) - 0.5) * 0.7, -torch.ones(n, 1)], 1)
    d = torch.nn.functional.normalize(d, dim=-1)
    gt = dict(rgb=torch.rand(n, 3, generator=g), sem=torch.randint(0, 6, (n,), generator=g),
              inst=torch.randint(0, 200, (n,), generator=g))
    return pagnerf_amd.Rays(o.to(dev), d.to(dev), dist_min=0.0, dist_max=1.9), {k: v.to(dev) for k, v in gt.items()}


def make_optimizer(nef):
    grid_params = [p for n, p in nef.named_parameters() if "grid" in n]
    rest = [p for n, p in nef.named_parameters() if "grid" not in n]
    groups = [dict(params=grid_params, lr=1e-3 * 100), dict(params=rest, lr=1e-3)]     # best.yaml:103,108 ; trainer.py:272-281
    try:
        return torch.optim.Adam(groups, eps=1e-15, fused=True)                         # config_parser.py:672
    except Exception:
        return torch.optim.Adam(groups, eps=1e-15)


def train_step(nef, tracer, opt, rays, gt, channels, world, sync=None):
    opt.zero_grad(set_to_none=True)
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


def cpu_baseline(n_rays, n_samples, budget_s=25.0):
    """The oracle's torch-CPU restatement of the reference's hash_grid_torch path, forward + backward."""
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
    if t_w > budget_s:                       # pathological host: report the single step rather than overrun
        return dict(value=n_rays / t_w, unit="rays/s", cores=torch.get_num_threads(), kind="port",
                    sample="%d rays x %d samples, 1 step of %.1f s (warm-up only; host too slow for more)" % (n_rays, n_samples, t_w))
    t0, n = time.perf_counter(), 0
    while True:
        step()
        n += 1
        if time.perf_counter() - t0 > budget_s or n >= 50:
            break
    dt = (time.perf_counter() - t0) / n
    return dict(value=n_rays / dt, unit="rays/s", cores=torch.get_num_threads(), kind="port",
                sample="%d rays x %d samples, hash grid L=16 T=2^19 (grids/hash_grid_torch.py restated op for op), density+colour "
                       "decoders, compositing, rgb L1 loss, forward+backward, %d timed steps of %.2f s; oracle/ torch-CPU" %
                       (n_rays, n_samples, n, dt))


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # PAG_BENCH_SHARE_GPU=1 (testing only): every rank uses cuda:0 and the collectives go through gloo, so the multi-process
    # code path can be exercised on a one-GPU box; the real launch is one rank per GPU over RCCL ("nccl" on ROCm).
    share = os.environ.get("PAG_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    if world > 1:
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if world != args.gpus and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()
    from pagnerf_amd import ops

    nef, tracer = make_model(args, dev, seed=0)                 # same seed everywhere: replicated parameters
    rays, gt = make_rays(args.rays, dev, seed=1000 + rank)      # per-rank ray shard
    opt = make_optimizer(nef)
    sync = None
    if world > 1:
        from pagnerf_amd import shard
        early = [nef.delta_grid.tables] if hasattr(nef, "delta_grid") else []
        sync = shard.GradSync(list(nef.parameters()), early=early)
    channels = {"rgb", "depth", "semantics", "inst_embedding"} if args.channels == "all" else {"rgb"}

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    enc_name = "pag_%s_encode_fwd" % args.grid

    def timed(n_steps, chans, profile=False):
        barrier()
        if profile:       # HIP events around the roofline kernel only; the full per-entry-point breakdown comes from a separate pass
            ops.profile_start(only=None if os.environ.get("PAG_BENCH_PROFILE_ALL") else {enc_name})
        t0 = time.perf_counter()
        for _ in range(n_steps):
            train_step(nef, tracer, opt, rays, gt, chans, world, sync)
        barrier()
        dt = time.perf_counter() - t0
        prof = ops.profile_stop() if profile else None
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, prof

    for _ in range(args.warmup):
        train_step(nef, tracer, opt, rays, gt, channels, world, sync)
    dt, prof = timed(args.steps, channels, profile=True)

    M = args.rays * args.samples
    L_, F_ = (24, 2) if args.grid == "permuto" else (16, 2)
    verts = 4 if args.grid == "permuto" else 8
    out_bytes = 2 if args.precision == "bf16" else 4
    bytes_per_sample = 12 + L_ * verts * F_ * 4 + L_ * F_ * out_bytes       # xyz + table gathers + feature row
    # HBM bytes per launch of that kernel from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected separately, KB
    # units, FETCH_SIZE x2 on gfx950 as MI355X_MICROARCH.md prescribes) - measured offline on this exact configuration and
    # committed under profiles/; null for any other configuration.
    traffic = None
    cands = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_traffic_per_launch.json"))
    tf = os.path.join(ROOT, "profiles", cands[-1]) if cands else ""      # newest committed PMC pass (profiles/README.md)
    if os.path.exists(tf) and (args.grid, args.rays, args.samples, args.precision) == ("permuto", 4096, 512, "bf16"):
        for k, v in json.load(open(tf)).items():
            if "permuto_fwd_kernel" in k:
                traffic = v["hbm_bytes_per_launch_corrected"]
    enc_ms = prof.get(enc_name, [])
    roofline = None
    if enc_ms:
        mean_ms = float(np.mean(enc_ms))
        achieved = bytes_per_sample * M / (mean_ms * 1e-3) / 1e9
        roofline = dict(bound="hbm", kernel=enc_name.replace("pag_", "") + "_kernel", achieved=round(achieved, 1),
                        peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic,
                        launches=len(enc_ms), avg_launch_ms=round(mean_ms, 4),
                        algorithmic_bytes_per_launch=bytes_per_sample * M)
    # per-entry-point device time: a separate, untimed pass with events around every C-ABI call
    n_bd = max(1, min(5, args.steps))
    barrier()
    ops.profile_start()
    for _ in range(n_bd):
        train_step(nef, tracer, opt, rays, gt, channels, world, sync)
    prof_all = ops.profile_stop()
    breakdown = {k.replace("pag_", ""): dict(calls_per_step=len(v) / n_bd, ms_per_step=round(float(np.sum(v)) / n_bd, 4))
                 for k, v in sorted(prof_all.items())}

    aux = None
    if not args.no_aux and args.channels == "all":
        for _ in range(2):
            train_step(nef, tracer, opt, rays, gt, {"rgb"}, world, sync)
        dt_rgb, _ = timed(max(3, args.steps // 2), {"rgb"})
        n_aux = max(3, args.steps // 2)
        aux = dict(workload="same scene, channels {rgb} only (epochs < 601, best.yaml:89)",
                   value=round(world * args.rays * n_aux / dt_rgb, 1), unit="rays/s", ms_per_step=round(dt_rgb / n_aux * 1e3, 3))

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(64, args.samples)

    if rank == 0:
        line = dict(
            metric="rays/sec (train step) on BUP20-shape scene", value=round(world * args.rays * args.steps / dt, 1), unit="rays/s",
            n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(dt / args.steps * 1e3, 3),
            higher_is_better=True, scaling="weak", vs_baseline=None, dtype=args.precision, data="synthetic",
            config=dict(workload="BUP20-shaped single view, PanopticDeltaNeF + %s grid (main+delta), %d rays x %d samples per GPU "
                                 "(M=%d packed samples), channels %s, train step fwd+bwd+Adam%s" %
                                 ("permutohedral L=24 F=2 T=2^18" if args.grid == "permuto" else "hash L=16 F=2 T=2^19", args.rays,
                                  args.samples, M, "+".join(sorted(channels)), ", RCCL grad all-reduce" if world > 1 else ""),
                        rays_per_gpu=args.rays, samples_per_ray=args.samples, grid=args.grid, channels=sorted(channels),
                        parallelism="ray-sharded data parallel x%d" % world),
            roofline=roofline, cpu_baseline=cpu, kernel_ms_per_step=breakdown, rgb_only=aux)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
