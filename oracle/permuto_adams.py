"""Second, independent float64 statement of the permutohedral-lattice encoding - written from the
DEFINITIONS in Adams, Baek, Davis 2010 ("Fast high-dimensional filtering using the permutohedral
lattice", sections 3.1-3.3) rather than from the incremental formulas oracle/permuto_encode.py (and
the HIP kernel written against it) use.

TEST INFRASTRUCTURE - see oracle/__init__.py.  PARITY UNPINNED like permuto_encode.py: the
third-party package the reference calls (grids/permuto_grid.py:57-62,71) is absent.  What this
file buys is that a misreading of the published algorithm shared by the oracle and the kernel is no
longer invisible: nothing below shares code or a formula with them.

  definition used here                                   vs. oracle/permuto_encode.py
  ------------------------------------------------------------------------------------------------
  elevation   explicit (d+1) x d matrix product           running prefix sums
  rank        np.argsort of the residuals, recomputed     pairwise comparison counts, then a
              from scratch AFTER the remainder-0 point     `rank + s` wrap
              has been moved onto the hyperplane
  vertices    rem0 + canonical-simplex table row           closed form rem0 + r - 4*[rank > 3-r]
              c_k = (k,..,k, k-(d+1),..,k-(d+1))
  weights     solve  sum_k b_k v_k = E, sum_k b_k = 1      +delta / -delta slot updates
              as a linear system per point
  hash        exact integer arithmetic mod 2^32 on int64   uint32 wrap-around of numpy casts

Constants that both share, because the call site / SURVEY Appendix B fix them: the per-axis scale
1/(sqrt((i+1)(i+2)) * scale[l]), the per-level shift, the multiplier 2531011, `% capacity`.
"""
import numpy as np

D = 3
HASH_MUL = 2531011


def elevation_matrix(d=D):
    """[d+1, d]: column i is the i-th basis vector of the hyperplane sum(x) = 0 (Adams 2010 eq. for E, unnormalised -
    the 1/sqrt((i+1)(i+2)) normalisation lives in the scale factors): rows 0..i hold +1, row i+1 holds -(i+1)."""
    E = np.zeros((d + 1, d))
    for i in range(d):
        E[:i + 1, i] = 1.0
        E[i + 1, i] = -(i + 1.0)
    return E


def canonical_simplex(d=D):
    """[d+1, d+1]: row k = vertex k of the canonical simplex in SORTED coordinate order:
    (k, ..., k, k-(d+1), ..., k-(d+1)) with d+1-k leading entries (Adams 2010 section 3.1)."""
    c = np.zeros((d + 1, d + 1), dtype=np.int64)
    for k in range(d + 1):
        c[k, :d + 1 - k] = k
        c[k, d + 1 - k:] = k - (d + 1)
    return c


def enclosing_simplex(xyz, shift_l, sf_l):
    """One level.  xyz [M,3] -> (vertices int64 [M,4,4] (vertex k has remainder k), bary f64 [M,4], E f64 [M,4])."""
    x = np.asarray(xyz, dtype=np.float64)
    cf = (x + np.asarray(shift_l, np.float64)) * np.asarray(sf_l, np.float64)
    E = cf @ elevation_matrix().T                                       # [M,4], rows sum to 0
    # nearest remainder-0 point: round every coordinate to the nearest multiple of d+1 ...
    # ("up if strictly closer to up, else down": exact ties have measure zero)
    down = np.floor(E / (D + 1)).astype(np.int64) * (D + 1)
    up = np.ceil(E / (D + 1)).astype(np.int64) * (D + 1)
    r0 = np.where((up - E) < (E - down), up, down)
    # ... then walk back onto the hyperplane: sum(r0) = s*(d+1); the s coordinates whose residual is smallest step down
    # (s > 0) or the |s| with the largest residual step up (s < 0)  (Adams 2010 section 3.2, "rounding")
    s = r0.sum(1) // (D + 1)
    resid = E - r0
    order = np.argsort(-resid, axis=1, kind="stable")                  # order[:, 0] = coordinate with the largest residual
    M = x.shape[0]
    rows = np.arange(M)
    for j in range(D + 1):
        pos_from_small = D - j                                          # j-th largest = (D-j)-th smallest
        coord = order[:, j]
        step_dn = (s > 0) & (pos_from_small < s)
        step_up = (s < 0) & (j < -s)
        r0[rows, coord] += np.where(step_up, D + 1, 0) - np.where(step_dn, D + 1, 0)
    assert (r0.sum(1) == 0).all()
    # rank of every coordinate's residual in the NEW frame (0 = largest), straight from a sort
    resid = E - r0
    order = np.argsort(-resid, axis=1, kind="stable")
    rank = np.empty_like(order)
    rank[rows[:, None], order] = np.arange(D + 1)[None, :]
    can = canonical_simplex()
    verts = r0[:, None, :] + can[:, rank].transpose(1, 0, 2)            # [M, k, coord] = r0 + c_k[rank[coord]]
    # barycentric weights: 4 unknowns, 5 equations (4 coordinates + partition of unity), consistent because sum(E) = 0.
    # Solved in coordinates relative to r0 (weights are translation invariant): |E| reaches 1e6 on the finest levels.
    local = (verts - r0[:, None, :]).transpose(0, 2, 1).astype(np.float64)                               # [M,coord,k]
    A = np.concatenate([local, np.ones((M, 1, D + 1))], axis=1)                                          # [M,5,4]
    rhs = np.concatenate([E - r0, np.ones((M, 1))], axis=1)[:, :, None]                                  # [M,5,1]
    AtA = A.transpose(0, 2, 1) @ A
    Atb = A.transpose(0, 2, 1) @ rhs
    bary = np.linalg.solve(AtA, Atb)[:, :, 0]
    return verts, bary, E


def lattice_hash(keys, capacity):
    """keys int64 [..., 3] (the first d coordinates of a lattice point) -> row.  k = 0; k = (k + key_i) * 2531011 mod 2^32."""
    k = np.zeros(keys.shape[:-1], dtype=np.int64)
    for i in range(D):
        k = np.mod((k + keys[..., i]) % (1 << 32) * HASH_MUL, 1 << 32)
    return np.mod(k, capacity)


def lattice_hash_scalar(key, capacity):
    """The same in arbitrary-precision Python integers (for hand-checked vectors)."""
    k = 0
    for i in range(D):
        k = ((k + int(key[i])) * HASH_MUL) % (1 << 32)
    return k % capacity


def encode(xyz, tables, shifts, sf, chunk=200000):
    """f64 features [M, L*F] with tables [L,T,F]; also returns (idx int64 [L,M,4], bary f64 [L,M,4])."""
    tables = np.asarray(tables, dtype=np.float64)
    L, T, F = tables.shape
    M = len(xyz)
    out = np.empty((M, L * F))
    idx_all = np.empty((L, M, 4), dtype=np.int64)
    b_all = np.empty((L, M, 4))
    for l in range(L):
        for lo in range(0, M, chunk):
            hi = min(M, lo + chunk)
            verts, bary, _ = enclosing_simplex(xyz[lo:hi], shifts[l], sf[l])
            idx = lattice_hash(verts[:, :, :D], T)
            out[lo:hi, l * F:(l + 1) * F] = (tables[l][idx] * bary[:, :, None]).sum(1)
            idx_all[l, lo:hi] = idx
            b_all[l, lo:hi] = bary
    return out, idx_all, b_all


def unelevate(E, shift_l, sf_l):
    """Inverse of the elevation on the hyperplane: E f64 [M,4] -> xyz f64 [M,3] (columns of the matrix are orthogonal)."""
    Em = elevation_matrix()
    cf = E @ Em / (Em * Em).sum(0)
    return cf / np.asarray(sf_l, np.float64) - np.asarray(shift_l, np.float64)
