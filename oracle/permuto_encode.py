"""Oracle: permutohedral-lattice hash encoding, numpy fp32.

TEST INFRASTRUCTURE - see oracle/__init__.py.

PARITY UNPINNED.  The reference delegates this to the third-party CUDA package
``permutohedral_encoding`` (RaduAlexandru/permutohedral_encoding), which is not
vendored under the reference tree, is pinned to no version (README.md:45) and is
not installed here.  What the reference itself fixes is only the call site:

  grids/permuto_grid.py:53      scales = np.geomspace(coarsest, finest, num_lods)
  grids/permuto_grid.py:57-62   PermutoEncoding(3, capacity, num_lods, feature_dim, scales)
  grids/permuto_grid.py:65-71   interpolate(): coords -> fp16 -> float -> encoder, out [M, L*F]

The algorithm below restates the published method (Adams, Baek, Davis 2010,
"Fast high-dimensional filtering using the permutohedral lattice"; Rosu & Behnke
2023, "PermutoSDF") in the form SURVEY.md Appendix B records it, and is the
definition of record for this build: the HIP kernels are checked against THIS.

Per (point p, level l), d = 3:
  cf_i   = (p_i + shift[l][i]) * scale_factor[l][i],  scale_factor[l][i] = 1/(sqrt((i+1)(i+2)) * scale[l])
  E      = elevate(cf)            (E in R^4, sum E = 0)
  rem0   = nearest remainder-0 lattice point of E (per-coordinate rounding to multiples of 4)
  rank   = order of the residuals E - rem0 (descending), corrected by sum(rem0)/4
  b[0..3]= barycentric weights of the enclosing simplex (sum = 1)
  vertex r: key_i = rem0_i + r - (rank_i > 3 - r ? 4 : 0), i < 3
  hash   : k = 0; for i<3: k = (k + key_i) * 2531011 (uint32 wrap);  idx = k % capacity
  out[l*F+f] = sum_r b[r] * table[l][idx_r][f]

All fp32 operations are written as separate roundings (no fused multiply-add) in
a fixed order so a device kernel using the same order reproduces them bit for bit.
"""
import numpy as np

HASH_MUL = np.uint32(2531011)
F32 = np.float32


def scale_factors(scales):
    """[L] scales (grids/permuto_grid.py:53) -> f32 [L,3] per-axis factors."""
    scales = np.asarray(scales, dtype=np.float64)
    sf = np.empty((len(scales), 3), dtype=np.float64)
    for i in range(3):
        sf[:, i] = 1.0 / (np.sqrt((i + 1) * (i + 2)) * scales)
    return sf.astype(np.float32)


def lattice_simplex(xyz, shift_l, sf_l):
    """One level.  xyz f32 [M,3] -> (rem0 int32 [M,4], rank int32 [M,4], bary f32 [M,4])."""
    M = xyz.shape[0]
    cf = ((xyz + shift_l[None, :]).astype(F32) * sf_l[None, :]).astype(F32)          # [M,3]
    E = np.empty((M, 4), dtype=F32)
    sm = np.zeros(M, dtype=F32)
    for i in (3, 2, 1):
        E[:, i] = (sm - (F32(i) * cf[:, i - 1]).astype(F32)).astype(F32)
        sm = (sm + cf[:, i - 1]).astype(F32)
    E[:, 0] = sm

    v = (E * F32(0.25)).astype(F32)
    up = (np.ceil(v) * F32(4.0)).astype(F32)
    dn = (np.floor(v) * F32(4.0)).astype(F32)
    take_up = (up - E).astype(F32) < (E - dn).astype(F32)
    rem0 = np.where(take_up, up, dn).astype(np.int32)
    s = rem0.sum(axis=1) // 4                                # exact: sum is a multiple of 4

    resid = (E - rem0.astype(F32)).astype(F32)
    rank = np.zeros((M, 4), dtype=np.int32)
    for i in range(3):
        for j in range(i + 1, 4):
            lt = resid[:, i] < resid[:, j]
            rank[:, i] += lt
            rank[:, j] += ~lt
    rank = rank + s[:, None]
    low = rank < 0
    high = rank > 3
    rank = np.where(low, rank + 4, np.where(high, rank - 4, rank))
    rem0 = np.where(low, rem0 + 4, np.where(high, rem0 - 4, rem0)).astype(np.int32)

    bary = np.zeros((M, 5), dtype=F32)
    rows = np.arange(M)
    for i in range(4):
        delta = ((E[:, i] - rem0[:, i].astype(F32)).astype(F32) * F32(0.25)).astype(F32)
        a = 3 - rank[:, i]
        bary[rows, a] = (bary[rows, a] + delta).astype(F32)
        bary[rows, a + 1] = (bary[rows, a + 1] - delta).astype(F32)
    bary[:, 0] = (bary[:, 0] + (F32(1.0) + bary[:, 4]).astype(F32)).astype(F32)
    return rem0, rank.astype(np.int32), bary[:, :4]


def vertex_indices(rem0, rank, capacity):
    """-> int32 [M,4] table rows of the 4 simplex vertices."""
    M = rem0.shape[0]
    idx = np.empty((M, 4), dtype=np.int32)
    for r in range(4):
        k = np.zeros(M, dtype=np.uint32)
        for i in range(3):
            key = rem0[:, i] + r - np.where(rank[:, i] > 3 - r, 4, 0)
            k = (k + key.astype(np.int32).astype(np.uint32)) * HASH_MUL
        idx[:, r] = (k % np.uint32(capacity)).astype(np.int32)
    return idx


def permuto_encode(xyz, tables, shifts, sf):
    """xyz f32 [M,3]; tables f32 [L,T,F]; shifts f32 [L,3]; sf f32 [L,3]
    -> (feats f32 [M,L*F], idx int32 [L,M,4], bary f32 [L,M,4])."""
    xyz = np.ascontiguousarray(xyz, dtype=F32)
    tables = np.asarray(tables, dtype=F32)
    L, T, F = tables.shape
    M = xyz.shape[0]
    out = np.empty((M, L * F), dtype=F32)
    all_idx = np.empty((L, M, 4), dtype=np.int32)
    all_b = np.empty((L, M, 4), dtype=F32)
    for l in range(L):
        rem0, rank, bary = lattice_simplex(xyz, shifts[l].astype(F32), sf[l].astype(F32))
        idx = vertex_indices(rem0, rank, T)
        acc = np.zeros((M, F), dtype=F32)
        for r in range(4):
            acc = (acc + (tables[l][idx[:, r]] * bary[:, r:r + 1]).astype(F32)).astype(F32)
        out[:, l * F:(l + 1) * F] = acc
        all_idx[l] = idx
        all_b[l] = bary
    return out, all_idx, all_b


def permuto_encode_bwd(xyz, grad_out, T, shifts, sf):
    """grad_out f32 [M,L*F] -> grad_tables f32 [L,T,F] (fp64 accumulation)."""
    xyz = np.ascontiguousarray(xyz, dtype=F32)
    L = shifts.shape[0]
    F = grad_out.shape[1] // L
    g = np.zeros((L, T, F), dtype=np.float64)
    for l in range(L):
        rem0, rank, bary = lattice_simplex(xyz, shifts[l].astype(F32), sf[l].astype(F32))
        idx = vertex_indices(rem0, rank, T)
        go = grad_out[:, l * F:(l + 1) * F].astype(np.float64)
        for r in range(4):
            np.add.at(g[l], idx[:, r], go * bary[:, r:r + 1].astype(np.float64))
    return g.astype(F32)


def permuto_encode_bwd_xyz(xyz, tables, grad_out, shifts, sf):
    """d loss / d xyz f64->f32 [M,3].  Inside a simplex the barycentric weights are affine in the elevated point E:
    coordinate a adds +delta_a to bary[3-rank_a] and -delta_a to bary[4-rank_a] (bary[4] folds into bary[0]),
    delta_a = (E_a - rem0_a)/4, and E = (cf0+cf1+cf2, cf1+cf2-cf0, cf2-2cf1, -3cf2), cf = (x + shift)*sf
    (lattice_simplex above).  PARITY UNPINNED like the forward (third-party position gradient of
    permutohedral_encoding); tests check this closed form against central differences of a float64 restatement."""
    xyz = np.ascontiguousarray(xyz, dtype=F32)
    tables = np.asarray(tables, dtype=np.float64)
    L, T, F = tables.shape
    M = xyz.shape[0]
    dx = np.zeros((M, 3), dtype=np.float64)
    rows = np.arange(M)
    for l in range(L):
        rem0, rank, _ = lattice_simplex(xyz, shifts[l].astype(F32), sf[l].astype(F32))
        idx = vertex_indices(rem0, rank, T)
        go = grad_out[:, l * F:(l + 1) * F].astype(np.float64)
        gb = np.zeros((M, 5))
        for r in range(4):
            gb[:, r] = (tables[l][idx[:, r]] * go).sum(1)
        gb[:, 4] = gb[:, 0]
        gE = np.empty((M, 4))
        for a in range(4):
            slot = 3 - rank[:, a]
            gE[:, a] = 0.25 * (gb[rows, slot] - gb[rows, slot + 1])
        s = sf[l].astype(np.float64)
        dx[:, 0] += (gE[:, 0] - gE[:, 1]) * s[0]
        dx[:, 1] += (gE[:, 0] + gE[:, 1] - 2.0 * gE[:, 2]) * s[1]
        dx[:, 2] += (gE[:, 0] + gE[:, 1] + gE[:, 2] - 3.0 * gE[:, 3]) * s[2]
    return dx.astype(F32)


def permuto_encode_f64(xyz, tables, shifts, sf):
    """float64 restatement of permuto_encode (no fp32 rounding) for finite-difference checks of the position
    gradient; not bit-comparable with the fp32 definition near simplex faces."""
    x = np.asarray(xyz, dtype=np.float64)
    tables = np.asarray(tables, dtype=np.float64)
    L, T, F = tables.shape
    M = x.shape[0]
    out = np.empty((M, L * F))
    rows = np.arange(M)
    for l in range(L):
        cf = (x + shifts[l].astype(np.float64)) * sf[l].astype(np.float64)
        E = np.stack([cf[:, 0] + cf[:, 1] + cf[:, 2], cf[:, 2] + cf[:, 1] - cf[:, 0], cf[:, 2] - 2 * cf[:, 1], -3 * cf[:, 2]], 1)
        v = E * 0.25
        up, dn = np.ceil(v) * 4, np.floor(v) * 4
        rem0 = np.where((up - E) < (E - dn), up, dn).astype(np.int64)
        s = rem0.sum(1) // 4
        resid = E - rem0
        rank = np.zeros((M, 4), dtype=np.int64)
        for i in range(3):
            for j in range(i + 1, 4):
                lt = resid[:, i] < resid[:, j]
                rank[:, i] += lt
                rank[:, j] += ~lt
        rank += s[:, None]
        low, high = rank < 0, rank > 3
        rank = np.where(low, rank + 4, np.where(high, rank - 4, rank))
        rem0 = np.where(low, rem0 + 4, np.where(high, rem0 - 4, rem0))
        bary = np.zeros((M, 5))
        for i in range(4):
            delta = (E[:, i] - rem0[:, i]) * 0.25
            a = 3 - rank[:, i]
            np.add.at(bary, (rows, a), delta)
            np.add.at(bary, (rows, a + 1), -delta)
        bary[:, 0] += 1.0 + bary[:, 4]
        idx = vertex_indices(rem0.astype(np.int32), rank.astype(np.int32), T)
        acc = np.zeros((M, F))
        for r in range(4):
            acc += tables[l][idx[:, r]] * bary[:, r:r + 1]
        out[:, l * F:(l + 1) * F] = acc
    return out


def half_round(xyz):
    """float(half(xyz)): what the encoder receives under the reference trainer's autocast - grids/permuto_grid.py:65
    `@torch.cuda.amp.custom_fwd(cast_inputs=torch.half)` then :71 `.type(torch.float)` (round-to-nearest-even, as torch)."""
    return np.asarray(xyz, dtype=F32).astype(np.float16).astype(F32)
