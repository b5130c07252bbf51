"""How much of a TRAINED scene's packed batch carries exactly-zero compositing weights?  (measure before building a tile early-out.)

Behind an opaque surface the transmittance exp(-sum tau) underflows to exactly 0, so w_m = T_m alpha_m, every per-sample gradient of the compositing backward
and every table-gradient entry of those samples are exactly 0: a 32-sample decoder tile (or a bin entry) whose weights are all 0 could be skipped bit-identically.
Trains the bench's model on scripts/train_synthetic.py's analytic sphere (4096 rays x 512 'ray'-mode samples, dense occupancy, --steps steps), then on held-out
rays reports: the fraction of samples with w == 0 / T == 0, the fraction of aligned 32-sample tiles whose weights are all 0 - overall, over the rays that hit the
sphere, and per tile position along the ray - and checks that d loss / d sigma and d loss / d rgb of the compositing backward are exactly 0 on those samples.
usage: python3 scripts/zero_weight_tiles.py [--steps 3000]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch

import bench
from train_synthetic import scene_rays

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--samples", type=int, default=512)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    import __graft_entry__ as ge
    ge.build()
    from pagnerf_amd import ops
    args = bench.parse(["--rays", str(a.rays), "--samples", str(a.samples)])
    nef, tracer = bench.make_model(args, dev, seed=0), bench.make_tracer(args)
    opt = bench.make_optimizer(nef)
    gen = torch.Generator().manual_seed(123)
    out = dict(scene="analytic sphere (scripts/train_synthetic.py), %d rays x %d samples, dense occupancy" % (a.rays, a.samples), checkpoints=[])
    done = 0
    for upto in sorted({300, 1000, a.steps}):
        for _ in range(done, upto):
            rays, gt = scene_rays(a.rays, gen, dev)
            bench.train_step(nef, tracer, opt, rays, gt, ["rgb", "semantics", "inst_embedding"], 1)
        done = upto
        g2 = torch.Generator().manual_seed(999)
        rays, gt = scene_rays(a.rays, g2, dev)
        ridx, pidx, samples, depths, deltas, boundary = nef.grid.raymarch(rays, level=None, num_samples=a.samples, raymarch_type="ray")
        _, ridx32, pack_start, ray_of_pack = nef.grid._pack_cache
        with torch.no_grad():
            f = nef(coords=samples, ridx=ridx32, ray_dirs=rays.dirs, channels={"rgb", "density"}, ray_packs=(pack_start, ray_of_pack))
        sigma = f["density"].reshape(-1).detach().clone().requires_grad_(True)
        rgb = f["rgb"].reshape(-1, 3).detach().clone().requires_grad_(True)
        alpha, hit, out_rgb, out_depth, w = ops.composite(sigma, rgb, deltas.reshape(-1), depths.reshape(-1), pack_start, ray_of_pack, a.rays, bg_white=True)
        (10.0 * torch.abs(out_rgb - gt["rgb"]).mean() + 0.01 * out_depth.sum()).backward()
        M = w.shape[0]
        assert M == a.rays * a.samples
        wz = (w == 0).reshape(a.rays, a.samples)
        tau = (sigma.detach() * deltas.reshape(-1)).reshape(a.rays, a.samples)
        T = torch.exp(-(torch.cumsum(tau, 1) - tau))
        tz = (T == 0)
        tiles = wz.reshape(a.rays, a.samples // 32, 32).all(-1)                  # [rays, 16]
        tiles_T = tz.reshape(a.rays, a.samples // 32, 32).all(-1)
        hits = gt["sem"] >= 0
        gz = (sigma.grad.reshape(a.rays, a.samples) == 0) & (rgb.grad.reshape(a.rays, a.samples, 3) == 0).all(-1)
        ent = dict(steps=upto, psnr_db=round(float(-10 * torch.log10(((out_rgb.detach() - gt["rgb"]) ** 2).mean())), 2),
                   samples_w_zero=round(float(wz.float().mean()), 4), samples_T_zero=round(float(tz.float().mean()), 4),
                   samples_w_zero_sigma_zero=round(float((wz & (sigma.detach().reshape(a.rays, a.samples) == 0)).float().mean()), 4),
                   tiles_all_w_zero=round(float(tiles.float().mean()), 4), tiles_all_T_zero=round(float(tiles_T.float().mean()), 4),
                   tiles_all_T_zero_on_hit_rays=round(float(tiles_T[hits].float().mean()), 4), hit_ray_fraction=round(float(hits.float().mean()), 4),
                   tiles_all_T_zero_by_position=[round(float(x), 3) for x in tiles_T.float().mean(0)],
                   grads_exactly_zero_where_T_zero=bool(gz[tz].all()) if bool(tz.any()) else None,
                   max_sigma=round(float(sigma.detach().max()), 1), median_sigma_inside=round(float(sigma.detach()[sigma.detach() > 1].median()) if bool((sigma.detach() > 1).any()) else 0.0, 1))
        out["checkpoints"].append(ent)
    print(json.dumps(out))
