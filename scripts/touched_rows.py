#!/usr/bin/env python3
"""Which table rows does one encode-backward launch touch?  (measure before building a per-level regime split.)

    python scripts/touched_rows.py [--rays 4096] [--samples 512] [--tile 4096] [--out profiles/r06_touched_rows.json]

Self-contained (torch only, CPU or GPU; does not import oracle/ or the HIP library): the permutohedral lattice lookup of
grids/permuto_grid.py:57-71 (scales geomspace(1, 1e-4, 24), T = 2^18, random per-level shift) on the bench scene's packed samples -
4096 rays x 512 'ray'-mode samples, consecutive samples of a ray adjacent in memory - and per level:

  rows_touched        distinct table rows one launch writes (of 2^18): what a sparse gradient exchange would carry, and whether a level is
                      "coarse" in the sense of fitting ONE workgroup's LDS for the whole launch
  per_tile_mean/max   distinct rows per tile of `--tile` consecutive samples (8 rays at 4096): what an LDS accumulator per workgroup would hold
  merged_per_sample   entries left per sample after merging runs of equal vertex id between adjacent samples (what bin_kernel's DPP merge emits)
  slots_per_tile_pow2 LDS slots an open-addressing accumulator would need at load factor <= 0.5
"""
import argparse
import json
import math
import os
import sys

import numpy as np
import torch

HASH_MUL = 2531011


def scale_factors(scales):
    return torch.stack([1.0 / (math.sqrt((i + 1) * (i + 2)) * scales) for i in range(3)], 1).float()


def vertex_rows(xyz, shift, sf, capacity):
    """xyz f32 [M,3] -> int64 [M,4] table rows of the enclosing simplex's vertices (the published lattice construction; fp32 op order as the
    kernels' - statistics only need the row ids, a last-bit difference at a cell boundary moves no count)."""
    cf = (xyz + shift[None]) * sf[None]
    M = xyz.shape[0]
    E = torch.empty(M, 4, dtype=torch.float32, device=xyz.device)
    sm = torch.zeros(M, dtype=torch.float32, device=xyz.device)
    for i in (3, 2, 1):
        E[:, i] = sm - float(i) * cf[:, i - 1]
        sm = sm + cf[:, i - 1]
    E[:, 0] = sm
    v = E * 0.25
    up, dn = torch.ceil(v) * 4.0, torch.floor(v) * 4.0
    rem0 = torch.where((up - E) < (E - dn), up, dn).to(torch.int64)
    s = rem0.sum(1) // 4
    resid = E - rem0.float()
    rank = torch.zeros(M, 4, dtype=torch.int64, device=xyz.device)
    for i in range(3):
        for j in range(i + 1, 4):
            lt = resid[:, i] < resid[:, j]
            rank[:, i] += lt
            rank[:, j] += ~lt
    rank = rank + s[:, None]
    low, high = rank < 0, rank > 3
    rem0 = torch.where(low, rem0 + 4, torch.where(high, rem0 - 4, rem0))
    rank = torch.where(low, rank + 4, torch.where(high, rank - 4, rank))
    rows = torch.empty(M, 4, dtype=torch.int64, device=xyz.device)
    for r in range(4):
        k = torch.zeros(M, dtype=torch.int64, device=xyz.device)
        for i in range(3):
            key = rem0[:, i] + r - torch.where(rank[:, i] > 3 - r, 4, 0)
            k = ((k + key) * HASH_MUL) & 0xFFFFFFFF
        rows[:, r] = k % capacity
    return rows


def bench_samples(n_rays, n_samples, seed, dev, occupancy=None):
    """The bench scene's rays (the geometry of bench.make_rays: downward-looking pinhole rays inside [-1,1]^3, near 0, far 1.9) marched as the
    'ray' mode does: depth = (linspace(0,1,S) + rand/S)^2 * (far - near) + near.  occupancy (bool [128^3] or None): keep the samples in kept cells."""
    g = torch.Generator().manual_seed(seed)
    o = torch.cat([(torch.rand(n_rays, 2, generator=g) - 0.5) * 0.6, torch.full((n_rays, 1), 0.95)], 1)
    d = torch.cat([(torch.rand(n_rays, 2, generator=g) - 0.5) * 0.7, -torch.ones(n_rays, 1)], 1)
    d = torch.nn.functional.normalize(d, dim=-1)
    near, far = 0.0, 1.9
    t = (torch.linspace(0, 1, n_samples)[None] + torch.rand(n_rays, n_samples, generator=g) / n_samples) ** 2 * (far - near) + near
    xyz = o[:, None] + t[..., None] * d[:, None]
    keep = (xyz.abs() <= 1.0).all(-1)
    if occupancy is not None:
        c = ((xyz + 1.0) * 64.0).floor().long().clamp(0, 127)
        keep &= occupancy[(c[..., 0] * 128 + c[..., 1]) * 128 + c[..., 2]]
    return xyz.to(dev), keep.to(dev)


def plant_row_mask(fraction, seed=0):
    """bench.synthetic_prune's occupancy: a 'plant row' blob covering `fraction` of the 128^3 cells."""
    R = 128
    ar = (torch.arange(R, dtype=torch.float32) + 0.5) / R * 2 - 1
    x, y, z = torch.meshgrid(ar, ar, ar, indexing="ij")
    gen = torch.Generator().manual_seed(seed)
    f = torch.zeros(R, R, R)
    for _ in range(24):
        c = torch.rand(3, generator=gen) * torch.tensor([1.6, 0.8, 0.8]) - torch.tensor([0.8, 0.4, 0.9])
        s_ = 0.08 + 0.2 * torch.rand(3, generator=gen)
        f += torch.exp(-(((x - c[0]) / s_[0]) ** 2 + ((y - c[1]) / s_[1]) ** 2 + ((z - c[2]) / s_[2]) ** 2))
    thr = torch.quantile(f.reshape(-1)[::7], 1.0 - fraction)
    return (f > thr).reshape(-1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--samples", type=int, default=512)
    ap.add_argument("--tile", type=int, default=4096)
    ap.add_argument("--levels", type=int, default=24)
    ap.add_argument("--log2-capacity", type=int, default=18)
    ap.add_argument("--seed", type=int, default=1000)
    ap.add_argument("--occupancy", type=float, default=None, help="post-prune regime: fraction of occupied 128^3 cells (bench.synthetic_prune's blob); samples outside are dropped")
    ap.add_argument("--ranks", type=int, default=1, help="also report the UNION of the rows touched by this many ranks' ray shards (seeds seed .. seed + ranks - 1)")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    T = 1 << a.log2_capacity
    scales = torch.from_numpy(np.geomspace(1.0, 1e-4, a.levels))
    sf = scale_factors(scales).to(dev)
    g = torch.Generator().manual_seed(0)
    shifts = (torch.rand(a.levels, 3, generator=g) * 10.0).to(dev)           # PermutoEncoding draws a random shift per level
    occ = plant_row_mask(a.occupancy) if a.occupancy else None
    xyz, keep = bench_samples(a.rays, a.samples, a.seed, dev, occ)
    xyz = xyz.half().float()                                                 # custom_fwd(cast_inputs=torch.half), permuto_grid.py:65
    pts = xyz[keep]                                                          # packed: rays in order, samples of a ray adjacent
    others = []
    for r in range(1, a.ranks):
        x2, k2 = bench_samples(a.rays, a.samples, a.seed + r, dev, occ)
        others.append(x2.half().float()[k2])
    M = pts.shape[0]
    n_tiles = (M + a.tile - 1) // a.tile
    levels = []
    for l in range(a.levels):
        rows = vertex_rows(pts, shifts[l], sf[l], T)                         # [M,4]
        uniq = torch.unique(rows)
        touched = int(uniq.numel())
        union = touched
        if others:
            hit = torch.zeros(T, dtype=torch.bool, device=dev)
            hit[uniq] = True
            for o_ in others:
                hit[vertex_rows(o_, shifts[l], sf[l], T).reshape(-1)] = True
            union = int(hit.sum())
        # run merge: vertex slot r of sample i merges with sample i-1 when ANY slot of i-1 holds the same row (bin_kernel merges per slot after
        # sorting the four ids; equality of the sorted tuples' members is what counts)
        prev = torch.cat([rows.new_full((1, 4), -1), rows[:-1]])
        same = (rows[:, :, None] == prev[:, None, :]).any(-1)
        merged = float((~same).float().sum() / M)
        tile_id = torch.arange(M, device=dev) // a.tile
        key = torch.unique((tile_id[:, None] * T + rows).reshape(-1))
        per_tile = torch.bincount(key // T, minlength=n_tiles).float()
        levels.append(dict(level=l, scale=float(scales[l]), rows_touched=touched, fill=round(touched / T, 5), union_rows=union, union_fill=round(union / T, 5),
                           merged_per_sample=round(merged, 4), per_tile_mean=round(float(per_tile.mean()), 1), per_tile_max=int(per_tile.max()),
                           slots_per_tile_pow2=int(2 ** math.ceil(math.log2(max(2.0 * float(per_tile.max()), 2.0))))))
        print("level %2d scale %.2e  rows touched %7d (%.3f) union of %d ranks %7d (%.3f)  per %d-sample tile: mean %8.1f max %6d  merged entries/sample %.3f" %
              (l, float(scales[l]), touched, touched / T, a.ranks, union, union / T, a.tile, float(per_tile.mean()), int(per_tile.max()), merged), flush=True)
    out = dict(rays=a.rays, samples_per_ray=a.samples, occupancy=a.occupancy, ranks=a.ranks, packed_samples=M, tile=a.tile, tiles=n_tiles, capacity=T, levels=levels,
               raw_entries_per_sample=4 * a.levels, merged_entries_per_sample=round(sum(x["merged_per_sample"] for x in levels), 2),
               union_fill_all_levels=round(sum(x["union_rows"] for x in levels) / (T * a.levels), 4),
               note="bench scene (bench.make_rays' downward pinhole rays, 'ray' march); torch restatement of the lattice lookup inside this script")
    if a.out:
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
