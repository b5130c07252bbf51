"""Oracle: multiresolution hash-grid encoding (Instant-NGP style), torch CPU.

TEST INFRASTRUCTURE - see oracle/__init__.py.

Restates, op for op (so that fp32 results are bit-identical on CPU), the
arithmetic of the reference's in-tree pure-torch encoder:

  grids/hash_grid_torch.py:13-24    hash()               xor of coord*prime, masked
  grids/hash_grid_torch.py:26-46    get_voxel_vertices() clamp, floor, 8 corners
  grids/hash_grid_torch.py:49-65    level growth factor b, resolutions floor(Nmin*b**i)
  grids/hash_grid_torch.py:69-93    trilinear_interp()   x then y then z lerps
  grids/hash_grid_torch.py:95-108   forward()            level-major concat

Quirks that are part of the contract:
  * every level is hashed (no dense coarse levels);
  * resolutions are derived in fp32 (16..2048 over 16 levels ends at 2047);
  * int32 wrapping multiply == uint32 arithmetic;
  * the clamp to [-1,1] is used for the cell lookup only, the interpolation
    weights use the unclamped point.
"""
import numpy as np
import torch

PRIMES = (1, 2654435761, 805459861)

# corner order of grids/hash_grid_torch.py:10 : i (x) outermost, k (z) innermost
CORNERS = np.array([[i, j, k] for i in (0, 1) for j in (0, 1) for k in (0, 1)], dtype=np.int32)


def level_resolutions(base_resolution, finest_resolution, n_levels):
    """fp32 resolutions as grids/hash_grid_torch.py:59,99 computes them."""
    base = torch.tensor(base_resolution)
    fine = torch.tensor(finest_resolution)
    b = torch.exp((torch.log(fine) - torch.log(base)) / (n_levels - 1))
    return [float(torch.floor(base * b ** i)) for i in range(n_levels)]


def corner_hash(corner_idx, log2_T):
    """corner_idx: int32 [...,3] -> int32 [...] in [0, 2^log2_T).  (hash_grid_torch.py:13-24)"""
    c = corner_idx.astype(np.uint32)
    h = (c[..., 0] * np.uint32(PRIMES[0])) ^ (c[..., 1] * np.uint32(PRIMES[1])) ^ (c[..., 2] * np.uint32(PRIMES[2]))
    return (h & np.uint32((1 << log2_T) - 1)).astype(np.int32)


def hash_level_indices(xyz, resolution, log2_T):
    """Per-level cell lookup. xyz torch f32 [M,3].  Returns (vmin, vmax, idx[M,8] int32)."""
    res = torch.tensor(float(resolution), dtype=torch.float32)
    lo = -torch.ones(3)
    hi = torch.ones(3)
    xc = torch.clamp(xyz, min=lo, max=hi)
    cell = (hi - lo) / res
    bl = torch.floor((xc - lo) / cell).int()
    vmin = bl * cell + lo
    vmax = vmin + torch.tensor([1.0, 1.0, 1.0]) * cell
    corners = bl.numpy()[:, None, :] + CORNERS[None]
    return vmin, vmax, corner_hash(corners, log2_T)


def hash_encode(xyz, tables, resolutions, log2_T):
    """xyz f32 [M,3]; tables f32 [L,T,F]; -> (feats f32 [M,L*F], idx int32 [L,M,8])."""
    xyz = xyz.float()
    outs, all_idx = [], []
    for lvl, res in enumerate(resolutions):
        vmin, vmax, idx = hash_level_indices(xyz, res, log2_T)
        emb = tables[lvl][torch.from_numpy(idx.astype(np.int64))]          # [M,8,F]
        w = (xyz - vmin) / (vmax - vmin)
        wx, wy, wz = w[:, 0:1], w[:, 1:2], w[:, 2:3]
        c00 = emb[:, 0] * (1 - wx) + emb[:, 4] * wx
        c01 = emb[:, 1] * (1 - wx) + emb[:, 5] * wx
        c10 = emb[:, 2] * (1 - wx) + emb[:, 6] * wx
        c11 = emb[:, 3] * (1 - wx) + emb[:, 7] * wx
        c0 = c00 * (1 - wy) + c10 * wy
        c1 = c01 * (1 - wy) + c11 * wy
        outs.append(c0 * (1 - wz) + c1 * wz)
        all_idx.append(idx)
    return torch.cat(outs, dim=-1), np.stack(all_idx)


def hash_encode_bwd(xyz, grad_out, n_entries, resolutions, log2_T):
    """d loss / d tables for hash_encode (what autograd through the reference gives).
    grad_out f32 [M,L*F] -> grad_tables f64-accumulated f32 [L,T,F]."""
    L = len(resolutions)
    F = grad_out.shape[1] // L
    xyz = xyz.float()
    g = torch.zeros(L, n_entries, F, dtype=torch.float64)
    for lvl, res in enumerate(resolutions):
        vmin, vmax, idx = hash_level_indices(xyz, res, log2_T)
        w = ((xyz - vmin) / (vmax - vmin)).double()
        go = grad_out[:, lvl * F:(lvl + 1) * F].double()
        for c in range(8):
            i, j, k = CORNERS[c]
            wc = (w[:, 0] if i else 1 - w[:, 0]) * (w[:, 1] if j else 1 - w[:, 1]) * (w[:, 2] if k else 1 - w[:, 2])
            g[lvl].index_add_(0, torch.from_numpy(idx[:, c].astype(np.int64)), go * wc[:, None])
    return g.float()


def hash_encode_bwd_xyz(xyz, tables, grad_out, resolutions, log2_T):
    """d loss / d xyz f32 [M,3]: autograd through hash_encode, exactly what the reference's autograd yields through
    grids/hash_grid_torch.py:69-108 (the cell lookup is integer, only the weights w = (x - vmin)/(vmax - vmin) carry
    a gradient; pinned by tests/golden/g7_hash_grad.npz)."""
    x = xyz.detach().float().clone().requires_grad_(True)
    out, _ = hash_encode(x, tables.detach(), resolutions, log2_T)
    out.backward(grad_out)
    return x.grad
