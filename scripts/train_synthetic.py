"""End-to-end training check on an analytic scene (no dataset in this image): a textured sphere of radius 0.5 seen by downward
cameras, white background.  Ground truth per ray is computed in closed form (ray-sphere intersection): colour from the hit
point, semantic class = quadrant of the hit point, instance id = longitude sector; rays that miss carry the label -100
(F.nll_loss's ignore_index: the composited class probabilities of an empty ray are alpha * sum = 0 by construction, as in the
reference, tracers/panoptic_packed_rf_tracer.py:197-205, so they cannot be supervised).
Trains the bench's model / optimizer / loss (bench.py, BUP20 hyper-parameters) for --steps steps on fresh random rays and reports
PSNR, semantic and instance accuracy on held-out rays, once on the bf16 MFMA path and once on the fp32 parity path (the one the
oracle tests pin), same seeds.  usage: python3 scripts/train_synthetic.py [--steps 600] [--rays 4096] [--samples 128]
"""
import argparse
import json
import math
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench


def scene_rays(n, gen, dev):
    import pagnerf_amd
    o = torch.cat([(torch.rand(n, 2, generator=gen) - 0.5) * 1.2, torch.full((n, 1), 0.95)], 1)
    tgt = torch.cat([(torch.rand(n, 2, generator=gen) - 0.5) * 1.4, torch.full((n, 1), -0.2)], 1)
    d = torch.nn.functional.normalize(tgt - o, dim=-1)
    # sphere |p| = 0.5
    b = (o * d).sum(-1)
    c = (o * o).sum(-1) - 0.25
    disc = b * b - c
    hit = disc > 0
    t = -b - torch.sqrt(disc.clamp_min(0))
    p = o + d * t[:, None]
    rgb = torch.where(hit[:, None], 0.5 + 0.5 * torch.sin(p * 9.0 + torch.tensor([0.0, 2.0, 4.0])), torch.ones(n, 3))
    sem = torch.where(hit, 1 + (p[:, 0] > 0).long() + 2 * (p[:, 1] > 0).long(), torch.full((n,), -100, dtype=torch.long))     # 1..4
    lon = torch.atan2(p[:, 1], p[:, 0])
    inst = torch.where(hit, 1 + ((lon + math.pi) / (2 * math.pi) * 12).long().clamp(0, 11), torch.full((n,), -100, dtype=torch.long))
    rays = pagnerf_amd.Rays(o.to(dev), d.to(dev), dist_min=0.0, dist_max=1.9)
    return rays, dict(rgb=rgb.to(dev), sem=sem.to(dev), inst=inst.to(dev))


def run(precision, a, dev):
    args = bench.parse(["--rays", str(a.rays), "--samples", str(a.samples), "--grid", a.grid, "--precision", precision])
    nef, tracer = bench.make_model(args, dev, seed=0), bench.make_tracer(args)
    opt = bench.make_optimizer(nef)
    chans = ["rgb", "semantics", "inst_embedding"]
    gen = torch.Generator().manual_seed(123)
    for step in range(a.steps):
        rays, gt = scene_rays(a.rays, gen, dev)
        loss = bench.train_step(nef, tracer, opt, rays, gt, chans, 1)
    gen = torch.Generator().manual_seed(999)
    rays, gt = scene_rays(4 * a.rays, gen, dev)
    with torch.no_grad():
        import pagnerf_amd
        rb = pagnerf_amd.batch_render(pagnerf_amd.Pipeline(nef, tracer), rays, channels=chans, render_batch=a.rays)
    mse = float(((rb.rgb - gt["rgb"]) ** 2).mean())
    vs_oracle = None
    if a.oracle_psnr:
        # "PSNR vs ref" anchored on the CPU oracle: the TRAINED parameters rendered by the HIP path and by the oracle chain (same samples,
        # same jitter) on a subset of the held-out rays; the helper lives under tests/ (nothing outside tests/ imports oracle/)
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
        from test_gpu_trajectory import hip_vs_oracle_render
        import pagnerf_amd
        n = a.oracle_psnr
        sub = pagnerf_amd.Rays(rays.origins[:n], rays.dirs[:n], rays.dist_min, rays.dist_max)
        r = hip_vs_oracle_render(nef, tracer, sub, a.samples)
        g = gt["rgb"][:n].float().cpu()
        vs_oracle = dict(rays=n, samples=r["samples"], psnr_hip_vs_oracle_db=r["psnr_hip_vs_oracle_db"],
                         psnr_hip_vs_gt_db=round(-10 * math.log10(float(((r["rgb_hip"] - g) ** 2).mean())), 2),
                         psnr_oracle_vs_gt_db=round(-10 * math.log10(float(((r["rgb_oracle"] - g) ** 2).mean())), 2),
                         sem_max_abs_diff=round(r["sem_max_abs_diff"], 5), inst_max_abs_diff=round(r["inst_max_abs_diff"], 5))
    return dict(precision=precision, vs_oracle=vs_oracle, final_loss=float(loss.detach()), psnr_db=round(-10 * math.log10(mse), 2),
                sem_acc=round(float((rb.semantics.argmax(-1) == gt["sem"])[gt["sem"] >= 0].float().mean()), 4),
                inst_acc=round(float((rb.inst_embedding.argmax(-1) == gt["inst"])[gt["inst"] >= 0].float().mean()), 4))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--samples", type=int, default=128)
    ap.add_argument("--grid", default="permuto")
    ap.add_argument("--oracle-psnr", type=int, default=0, metavar="RAYS",
                    help="also render RAYS held-out rays of the trained model with the CPU oracle (tests/test_gpu_trajectory.py) and report the "
                         "PSNR of the HIP render against it, next to both renders' PSNR against the ground truth")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    import __graft_entry__ as ge
    ge.build()
    out = [run(p, a, dev) for p in ("bf16", "fp32")]
    print(json.dumps(dict(scene="analytic sphere, %d steps x %d rays x %d samples, %s grid" % (a.steps, a.rays, a.samples, a.grid), runs=out)))
