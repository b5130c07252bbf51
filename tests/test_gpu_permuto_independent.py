"""Checks of the HIP permutohedral encoder that do NOT go through oracle/permuto_encode.py (the file the kernel was written
against): geometric properties of the simplex the kernel actually picks, hand-derived hash rows, continuity across simplex
faces, and agreement with the independent float64 statement oracle/permuto_adams.py.  Plus the fp16 coordinate rounding the
reference trains with (grids/permuto_grid.py:65,71), full-size table-gradient checks and the checkpoint round trip.

Need a real MI355X:  pytest -m gpu
"""
import itertools

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# k = 0; for i in 0..2: k = (k + key_i) * 2531011 mod 2^32, worked by hand in tests/test_oracle_golden.py (HAND_HASH)
HAND_HASH = {(0, 0, 0): 0, (1, 1, 1): 865908135, (2, 2, -2): 1721692226, (3, -1, -1): 2295759813}


def _ops():
    from pagnerf_amd import ops, _lib
    return ops, _lib


def test_hand_derived_hash_rows_on_the_hip_kernel(gpu_device):
    """x = (0.5, 0.5, 0.5), unit scale factors, no shift elevates to E = (1.5, 0.5, -0.5, -1.5): the centroid of the canonical
    simplex (0,0,0,0), (1,1,1,-3), (2,2,-2,-2), (3,-1,-1,-1) - weights 1/4 each, rows = the hand-derived hashes of the first
    three coordinates.  A table that is zero except for 1, 2, 4, 8 at those rows must interpolate to exactly 3.75."""
    ops, L = _ops()
    for cap in (1 << 18, 1000003):
        tab = torch.zeros(1, cap, 1)
        for v, key in zip((1.0, 2.0, 4.0, 8.0), ((0, 0, 0), (1, 1, 1), (2, 2, -2), (3, -1, -1))):
            tab[0, HAND_HASH[key] % cap, 0] = v
        spec = ops.permuto_spec(np.ones((1, 3), np.float32), np.zeros((1, 3), np.float32), cap, 1)
        x = torch.full((1, 3), 0.5, device=gpu_device)
        out = ops.encode(x, tab.to(gpu_device), spec)
        assert float(out[0, 0]) == 3.75
        # the same rows receive the gradient, 1/4 each
        t = tab.to(gpu_device).requires_grad_(True)
        ops.BWD_ALGO = "atomic"
        try:
            ops.encode(x, t, spec).backward(torch.ones(1, 1, device=gpu_device))
        finally:
            ops.BWD_ALGO = "binned"
        nz = torch.nonzero(t.grad[0, :, 0]).reshape(-1).cpu().tolist()
        assert sorted(nz) == sorted(HAND_HASH[k] % cap for k in HAND_HASH)
        assert torch.equal(t.grad[0, nz, 0].cpu(), torch.full((4,), 0.25))


def _lattice_points_near(lo, hi):
    """All points of the permutohedral lattice (integer 4-vectors, zero sum, all coordinates congruent mod 4) whose first
    three coordinates lie in [lo, hi] - by plain enumeration."""
    pts = []
    rng = range(lo, hi + 1)
    for a, b, c in itertools.product(rng, rng, rng):
        d = -(a + b + c)
        if (a - b) % 4 == 0 and (a - c) % 4 == 0 and (a - d) % 4 == 0:
            pts.append((a, b, c, d))
    return np.array(pts, dtype=np.int64)


def test_simplex_the_kernel_picks_is_a_lattice_simplex_containing_the_point(gpu_device):
    """For single samples, the table gradient of the HIP encoder with grad_out = 1 IS (row -> barycentric weight).  With the
    rows mapped back to lattice points through an enumerated reverse hash table (exact integer hash, no oracle code), check the
    geometry of Adams et al. 2010 directly: weights >= 0 and sum to 1; the four vertices have remainders 0,1,2,3 mod 4;
    consecutive vertices differ by a permutation of (1,1,1,-3) (i.e. they ARE a Delaunay cell of A*_3); and
    sum_k b_k v_k reproduces the elevated sample."""
    ops, L = _ops()
    from oracle import permuto_adams as pa
    dev = gpu_device
    cap = 1 << 22
    pts = _lattice_points_near(-36, 36)
    rows = pa.lattice_hash(pts[:, :3], cap)
    uniq, first, counts = np.unique(rows, return_index=True, return_counts=True)
    row_to_pt = {int(r): pts[i] for r, i, c in zip(uniq, first, counts) if c == 1}       # rows with ONE enumerated preimage
    rs = np.random.RandomState(3)
    ops.BWD_ALGO = "atomic"
    try:
        checked = 0
        for scale, shift_std in ((0.5, 0.3), (0.3, 0.7), (0.2, 0.5)):
            sf = np.array([[1.0 / (np.sqrt((i + 1) * (i + 2)) * scale) for i in range(3)]], dtype=np.float32)
            shift = (rs.standard_normal((1, 3)) * shift_std).astype(np.float32)
            spec = ops.permuto_spec(sf, shift, cap, 1)
            xs = rs.uniform(-1, 1, size=(96, 3)).astype(np.float32)
            E = ((xs.astype(np.float64) + shift[0]) * sf[0].astype(np.float64)) @ pa.elevation_matrix().T
            assert np.abs(E).max() < 31, "enumeration box too small for this scale"
            for i in range(len(xs)):
                t = torch.zeros(1, cap, 1, device=dev, requires_grad=True)
                ops.encode(torch.from_numpy(xs[i:i + 1]).to(dev), t, spec).backward(torch.ones(1, 1, device=dev))
                nz = torch.nonzero(t.grad[0, :, 0]).reshape(-1)
                w = t.grad[0, nz, 0].double().cpu().numpy()
                r = nz.cpu().numpy()
                assert 1 <= len(r) <= 4
                assert (w >= -1e-6).all() and abs(w.sum() - 1.0) < 1e-5, (w, w.sum())
                if len(r) < 4 or any(int(q) not in row_to_pt for q in r):
                    continue                                   # a zero weight (point on a face) or a row with two preimages: skip
                v = np.stack([row_to_pt[int(q)] for q in r])
                assert (v.sum(1) == 0).all()
                rem = v[:, 0] % 4
                assert sorted(rem.tolist()) == [0, 1, 2, 3]
                v, w = v[np.argsort(rem)], w[np.argsort(rem)]   # vertex k = remainder k
                for k in range(4):
                    step = v[(k + 1) % 4] - v[k]
                    assert sorted(step.tolist()) == [-3, 1, 1, 1], (k, step)
                assert np.abs((w[:, None] * v).sum(0) - E[i]).max() < 2e-5 * max(1.0, np.abs(E[i]).max())
                checked += 1
        assert checked > 200
    finally:
        ops.BWD_ALGO = "binned"


def test_features_are_continuous_across_simplex_faces(gpu_device):
    """The interpolant is continuous: along a straight segment that crosses many simplex faces (every level of the best.yaml
    configuration) the per-step change of the HIP features stays within the Lipschitz bound of a piecewise-linear interpolant
    plus the fp32 resolution of the elevated coordinates - a wrong vertex or weight at a face shows up as an O(|table|) jump."""
    ops, L = _ops()
    from pagnerf_amd import grids
    from oracle import permuto_adams as pa
    dev = gpu_device
    rs = np.random.RandomState(5)
    Lv, cap, F = 24, 1 << 18, 2
    sf = grids.PermutoGridHIP.scale_factors(np.geomspace(1.0, 1e-4, Lv))
    shifts = (rs.standard_normal((Lv, 3)) * 10).astype(np.float32)
    tab = torch.from_numpy(rs.standard_normal((Lv, cap, F)).astype(np.float32)).to(dev)
    spec = ops.permuto_spec(sf, shifts, cap, F)
    n = 200001
    tmax = float(tab.abs().max())
    for seg_len in (1.0, 0.01):
        a = rs.uniform(-0.5, 0.5, size=3)
        dirv = rs.standard_normal(3)
        dirv /= np.linalg.norm(dirv)
        tt = np.linspace(0.0, seg_len, n)
        x64 = a[None] + tt[:, None] * dirv[None]
        x = torch.from_numpy(x64.astype(np.float32)).to(dev)
        f = ops.encode(x, tab, spec).double().cpu().numpy()
        dx = np.abs(np.diff(x.double().cpu().numpy(), axis=0)).sum(1).max()             # actual fp32 step (L1)
        jumps = np.abs(np.diff(f, axis=0)).max(0).reshape(Lv, F).max(1)
        for l in range(Lv):
            _, _, E = pa.enclosing_simplex(x64[::20000], shifts[l], sf[l].numpy())
            ulp = float(np.spacing(np.float32(np.abs(E).max())))
            # |d b / d E| <= 1/2 per coordinate pair, |d E / d x| <= 3 sf_0: generous Lipschitz constant 8 sf_0 max|table|
            bound = (8.0 * float(sf[l, 0]) * dx + 8.0 * ulp) * tmax + 1e-5
            assert jumps[l] <= bound, (seg_len, l, jumps[l], bound)
    # sanity of the test itself: a deliberately wrong shift on ONE side of a plane does produce jumps beyond the bound
    x = torch.from_numpy((np.linspace(0, 1, n)[:, None] * np.array([[0.3, 0.2, 0.1]])).astype(np.float32)).to(dev)
    f0 = ops.encode(x, tab, spec)
    spec2 = ops.permuto_spec(sf, shifts + 1.7, cap, F)
    f1 = ops.encode(x, tab, spec2)
    mixed = torch.where((torch.arange(n, device=dev) < n // 2)[:, None], f0, f1).double().cpu().numpy()
    assert np.abs(np.diff(mixed[:, :2], axis=0)).max() > 0.05


def test_hip_features_match_the_independent_f64_statement(gpu_device):
    """HIP fp32 features vs oracle/permuto_adams.py (float64, written from the paper's definitions) at best.yaml sizes."""
    ops, L = _ops()
    from pagnerf_amd import grids
    from oracle import permuto_adams as pa
    dev = gpu_device
    rs = np.random.RandomState(9)
    Lv, cap, F = 24, 1 << 18, 2
    sf = grids.PermutoGridHIP.scale_factors(np.geomspace(1.0, 1e-4, Lv)).numpy()
    shifts = (rs.standard_normal((Lv, 3)) * 10).astype(np.float32)
    tab = rs.standard_normal((Lv, cap, F)).astype(np.float32)
    x = rs.uniform(-1, 1, size=(100000, 3)).astype(np.float32)
    for half in (False, True):
        spec = ops.permuto_spec(sf, shifts, cap, F, half_coords=half)
        got = ops.encode(torch.from_numpy(x).to(dev), torch.from_numpy(tab).to(dev), spec).double().cpu().numpy()
        xin = x.astype(np.float16).astype(np.float32) if half else x
        ref, _, _ = pa.encode(xin, tab, shifts, sf)
        for l in range(Lv):
            _, _, E = pa.enclosing_simplex(xin[:2000], shifts[l], sf[l])
            ulp = float(np.spacing(np.float32(np.abs(E).max())))
            err = np.abs(got[:, l * F:(l + 1) * F] - ref[:, l * F:(l + 1) * F]).max()
            assert err <= (8 * ulp + 1e-5) * np.abs(tab[l]).max(), (half, l, err, ulp)


def test_half_coords_flag_equals_rounding_the_input(gpu_device):
    """PAG_ENC_HALF_COORDS: forward, table gradient and position gradient with the flag == the same call on float(half(x))
    without it, bit for bit; and == the oracle on oracle.half_round(x).  Both encoders, strided and XCD8 layouts."""
    ops, L = _ops()
    from oracle import permuto_encode as op, hash_encode as oh
    dev = gpu_device
    rs = np.random.RandomState(21)
    M, Lv, cap, F = 5000, 24, 1 << 14, 2
    x = rs.uniform(-1, 1, size=(M, 3)).astype(np.float32)
    xh = op.half_round(x)
    assert not np.array_equal(x, xh) and np.array_equal(xh, torch.from_numpy(x).half().float().numpy())
    sf = op.scale_factors(np.geomspace(1.0, 1e-4, Lv))
    shifts = (rs.standard_normal((Lv, 3)) * 10).astype(np.float32)
    tab = rs.standard_normal((Lv, cap, F)).astype(np.float32)
    go = rs.standard_normal((M, Lv * F)).astype(np.float32)
    res = oh.level_resolutions(16, 2048, 16)
    htab = rs.standard_normal((16, cap, F)).astype(np.float32)
    specs = [
        (ops.permuto_spec(sf, shifts, cap, F, half_coords=True), ops.permuto_spec(sf, shifts, cap, F), tab, go),
        (ops.hash_spec(res, 14, F, half_coords=True), ops.hash_spec(res, 14, F), htab, np.ascontiguousarray(go[:, :32])),
    ]
    for s_half, s_plain, tb, g in specs:
        t = torch.from_numpy(tb).to(dev)
        outs = []
        for spec, xin in ((s_half, x), (s_plain, xh)):
            xx = torch.from_numpy(xin).to(dev).requires_grad_(True)
            tt = t.clone().requires_grad_(True)
            o = ops.encode(xx, tt, spec)
            o.backward(torch.from_numpy(g).to(dev))
            o8 = ops.encode(xx.detach(), tt.detach(), spec, layout="xcd8")
            outs.append((o.detach(), tt.grad, xx.grad, o8))
        for a, b in zip(*outs):
            assert torch.equal(a, b)
        assert not torch.equal(outs[0][0], ops.encode(torch.from_numpy(x).to(dev), t, s_plain))     # the flag does something
    ref, _, _ = op.permuto_encode(xh, tab, shifts, sf)
    got = ops.encode(torch.from_numpy(x).to(dev), torch.from_numpy(tab).to(dev), specs[0][0]).cpu().numpy()
    assert np.array_equal(got, ref)
    # through the grid class: on by default for the permutohedral grid (as the reference trains), off for HashGridTorch
    import pagnerf_amd
    g = pagnerf_amd.PermutoGridHIP(2, capacity_log_2=14, num_lods=Lv, finest_scale=1e-4, blas_level=3)
    g.init_from_scales(random_shift=torch.from_numpy(shifts), tables=torch.from_numpy(tab))
    g = g.to(dev)
    assert g.half_coords and np.array_equal(g.interpolate(torch.from_numpy(x).to(dev)[:, None]).detach().cpu().numpy(), ref)
    h = pagnerf_amd.HashGridHIP(2, codebook_bitwidth=14, blas_level=3)
    assert not h.half_coords
    # when the rounding applies (grids.rounds_coords): the reference casts only under the train step's autocast (trainer.py:429);
    # validate() / evaluate_metrics() run pipeline.eval() without autocast (trainer.py:630,944) and nef.prune() runs outside it.
    xg = torch.from_numpy(x).to(dev)[:, None]
    plain, _, _ = op.permuto_encode(x, tab, shifts, sf)
    g.eval()
    assert not g.rounds_coords() and np.array_equal(g.interpolate(xg).detach().cpu().numpy(), plain)
    with torch.autocast("cuda"):
        assert g.rounds_coords() and np.array_equal(g.interpolate(xg).detach().float().cpu().numpy(), ref)
    g.train()
    with g.fp32_coords():
        assert not g.rounds_coords() and np.array_equal(g.interpolate(xg).detach().cpu().numpy(), plain)
    assert g.rounds_coords()
    g.half_coords = "always"
    g.eval()
    assert np.array_equal(g.interpolate(xg).detach().cpu().numpy(), ref)
    g.half_coords = False
    g.train()
    assert np.array_equal(g.interpolate(xg).detach().cpu().numpy(), plain)


def _xcd8_to_strided(g8, Lv, F):
    """bf16 [8, M, 8] XCD-grouped tensor -> [M, L*F] with the same values."""
    from pagnerf_amd import ops
    M = g8.shape[1]
    cols = ops.xcd8_columns(Lv, F)
    flat = g8.permute(1, 0, 2).reshape(M, 64)
    out = torch.zeros(M, Lv * F, device=g8.device, dtype=g8.dtype)
    for p, c in enumerate(cols):
        if c >= 0:
            out[:, c] = flat[:, p]
    return out


@pytest.mark.parametrize("kind", ["permuto", "hash"])
def test_full_size_table_gradients_binned_vs_atomic_and_oracle(gpu_device, kind):
    """BASELINE configs[1] / configs[2] size (4096 rays x 512 samples = 2 097 152 samples, production bf16 XCD8 gradients):
    the binned two-pass backward (2.4 GB workspace, 12-bit slice keys, 26-bit packed floats, per-level fixed-point scale) against
    the per-vertex global-atomic kernel - two independent algorithms, the atomic one oracle-checked at small sizes - and against
    the CPU oracle on three levels of a 64 k-sample subset."""
    ops, L = _ops()
    from pagnerf_amd import grids
    dev = gpu_device
    gen = torch.Generator().manual_seed(17)
    N, S = 4096, 512
    M = N * S
    # ray-shaped sample positions: consecutive samples of a ray are close (that is what the run merge of the bin pass exploits)
    o = torch.cat([(torch.rand(N, 2, generator=gen) - 0.5) * 0.6, torch.full((N, 1), 0.95)], 1)
    d = torch.nn.functional.normalize(torch.cat([(torch.rand(N, 2, generator=gen) - 0.5) * 0.7, -torch.ones(N, 1)], 1), dim=-1)
    tv = (torch.linspace(0, 1, S)[None] + torch.rand(N, S, generator=gen) / S) ** 2 * 1.9
    x = (o[:, None] + d[:, None] * tv[..., None]).reshape(M, 3).to(dev)
    if kind == "permuto":
        Lv, F, rows = 24, 2, 1 << 18
        shifts = torch.randn(Lv, 3, generator=gen) * 10
        sfac = grids.PermutoGridHIP.scale_factors(np.geomspace(1.0, 1e-4, Lv))
        spec = ops.permuto_spec(sfac, shifts, rows, F, half_coords=True)
    else:
        Lv, F, rows = 16, 2, 1 << 19
        from oracle import hash_encode as oh
        res = oh.level_resolutions(16, 2048, Lv)
        spec = ops.hash_spec(res, 19, F)
    # heavy-tailed gradient (a few rays dominate, as with a real loss) in the production layout
    g8 = (torch.randn(8, M, 8, generator=gen) * torch.exp(torch.randn(1, M, 1, generator=gen))).to(dev).bfloat16()
    gs = _xcd8_to_strided(g8, Lv, F)
    gt_binned = torch.empty(Lv, rows, F, device=dev)
    ops._encode_bwd(spec, x, g8, None, gt_binned, overwrite=True)
    ops.BWD_ALGO = "atomic"
    try:
        gt_atomic = torch.zeros(Lv, rows, F, device=dev)
        ops._encode_bwd(spec, x, gs, None, gt_atomic)
    finally:
        ops.BWD_ALGO = "binned"
    torch.cuda.synchronize()
    for l in range(Lv):
        a, b = gt_binned[l].double(), gt_atomic[l].double()
        rel = float((a - b).norm() / b.norm())
        # the fp32 atomic sum itself carries ~eps * sqrt(adds per row) relative error on the coarse levels (millions of adds per row)
        assert rel < 2e-4, (kind, l, rel)
        assert float((a - b).abs().max()) <= 2e-4 * float(b.abs().max()) + 1e-6, (kind, l)
    # determinism of the binned path (integer accumulation): bitwise identical on a second run
    gt2 = torch.empty_like(gt_binned)
    ops._encode_bwd(spec, x, g8, None, gt2, overwrite=True)
    assert torch.equal(gt2, gt_binned)
    # oracle on a 64 k-sample subset (16 whole rays... 128 rays x 512), three levels
    sub = slice(0, 128 * S)
    xs = x[sub].contiguous()
    gsub8 = g8[:, sub].contiguous()
    gt_sub = torch.empty(Lv, rows, F, device=dev)
    ops._encode_bwd(spec, xs, gsub8, None, gt_sub, overwrite=True)
    gs_np = gs[sub].float().cpu().numpy()
    xs_np = xs.cpu().numpy()
    for l in ((0, 12, 23) if kind == "permuto" else (0, 8, 15)):
        if kind == "permuto":
            from oracle import permuto_encode as op
            ref = op.permuto_encode_bwd(op.half_round(xs_np), gs_np[:, l * F:(l + 1) * F], rows, shifts.numpy()[l:l + 1], sfac.numpy()[l:l + 1])[0]
        else:
            ref = oh.hash_encode_bwd(torch.from_numpy(xs_np), torch.from_numpy(gs_np[:, l * F:(l + 1) * F]), rows, res[l:l + 1], 19)[0].numpy()
        got = gt_sub[l].cpu().numpy()
        np.testing.assert_allclose(got, ref, rtol=2e-4, atol=2e-5 * float(np.abs(ref).max()), err_msg="%s level %d" % (kind, l))


def test_checkpoint_round_trip_renders_identically(gpu_device):
    """state_dict -> freshly seeded model -> identical render and identical sample count (pruned occupancy and
    per-level shifts included); the same through torch.save(pipeline) / torch.load (the reference's default format)."""
    import io
    import pagnerf_amd
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    import test_gpu_parity as T
    dev = gpu_device
    nef, tracer, rays, occ, jitter = T._make_scene(dev, "bf16", seed=0, N=256, S=64, cap_log2=12)
    chans = {"rgb", "depth", "semantics", "inst_embedding"}
    with torch.no_grad():
        rb = tracer(nef, channels=chans, rays=rays, jitter=jitter.to(dev))
    n_samples = nef.grid._pack_cache[0].shape[0]
    other, _, _, _, _ = T._make_scene(dev, "bf16", seed=7, N=8, S=8, cap_log2=12)
    for grid in (other.grid, other.delta_grid):
        grid.blas_init(torch.ones(32 ** 3, dtype=torch.bool))
    with torch.no_grad():
        before = tracer(other, channels=chans, rays=rays, jitter=jitter.to(dev))
    assert not torch.equal(before.rgb, rb.rgb)
    res = other.load_state_dict(nef.state_dict())
    assert not res.missing_keys and not res.unexpected_keys
    with torch.no_grad():
        again = tracer(other, channels=chans, rays=rays, jitter=jitter.to(dev))
    assert other.grid._pack_cache[0].shape[0] == n_samples
    for ch in ("rgb", "alpha", "depth", "semantics", "inst_embedding", "hit"):
        assert torch.equal(getattr(again, ch), getattr(rb, ch)), ch
    buf = io.BytesIO()
    torch.save(pagnerf_amd.Pipeline(nef, tracer), buf)
    buf.seek(0)
    pipe = torch.load(buf, weights_only=False)
    with torch.no_grad():
        third = pipe(channels=chans, rays=rays, jitter=jitter.to(dev))
    for ch in ("rgb", "alpha", "depth", "semantics", "inst_embedding", "hit"):
        assert torch.equal(getattr(third, ch), getattr(rb, ch)), ch


@pytest.mark.parametrize("kind", ["permuto", "hash"])
def test_binned_table_gradients_at_ragged_sizes(gpu_device, kind):
    """The reduce pass reads the (tile, slice) segments of up to 64 tiles as one stream, two 64-entry chunks in flight, with a three-compare
    tile search where segments are long and a binary search where they are short, and skips tile groups that hold nothing of its slice.
    Sizes that put every one of those branches at its edge: a single sample, one chunk exactly, fewer tiles than waves, a last group with
    one tile, tile counts that are not multiples of the group size, and whole-ray as well as shuffled (no repeats: every level
    'distinct') sample orders - binned == per-vertex atomics, and bitwise reproducible."""
    ops, L = _ops()
    from pagnerf_amd import grids
    dev = gpu_device
    gen = torch.Generator().manual_seed(5)
    if kind == "permuto":
        Lv, F, rows, ts = 24, 2, 1 << 18, 1024
        spec = ops.permuto_spec(grids.PermutoGridHIP.scale_factors(np.geomspace(1.0, 1e-4, Lv)), torch.randn(Lv, 3, generator=gen) * 10, rows, F,
                                half_coords=True)
    else:
        Lv, F, rows, ts = 16, 2, 1 << 19, 512
        from oracle import hash_encode as oh
        spec = ops.hash_spec(oh.level_resolutions(16, 2048, Lv), 19, F)
    for M, shuffled in ((1, False), (63, False), (64, True), (65, False), (ts - 1, False), (ts, True), (ts + 1, False), (3 * ts + 5, True),
                        (17 * ts + 1, False), (64 * ts, False), (65 * ts + 13, True), (130 * ts + 700, False)):
        n_rays = max(1, M // 200)
        o = (torch.rand(n_rays, 1, 3, generator=gen) - 0.5) * 0.8
        d = torch.nn.functional.normalize(torch.randn(n_rays, 1, 3, generator=gen), dim=-1)
        per = (M + n_rays - 1) // n_rays
        t = torch.linspace(0, 1, per)[None, :, None] * 0.9
        x = (o + d * t).reshape(-1, 3)[:M]
        if shuffled:
            x = x[torch.randperm(M, generator=gen)]
        x = x.contiguous().to(dev)
        g8 = (torch.randn(8, M, 8, generator=gen) * torch.exp(torch.randn(1, M, 1, generator=gen))).to(dev).bfloat16()
        gs = _xcd8_to_strided(g8, Lv, F)
        binned = torch.full((Lv, rows, F), float("nan"), device=dev)
        ops._encode_bwd(spec, x, g8, None, binned, overwrite=True)          # overwrite: every row of the table is written, NaNs must be gone
        ops.BWD_ALGO = "atomic"
        try:
            atomic = torch.zeros(Lv, rows, F, device=dev)
            ops._encode_bwd(spec, x, gs, None, atomic)
        finally:
            ops.BWD_ALGO = "binned"
        assert torch.isfinite(binned).all(), (kind, M)
        for l in range(Lv):
            a, b = binned[l].double(), atomic[l].double()
            assert float((a - b).abs().max()) <= 2e-4 * float(b.abs().max()) + 1e-7, (kind, M, shuffled, l)
        again = torch.empty_like(binned)
        ops._encode_bwd(spec, x, g8, None, again, overwrite=True)
        assert torch.equal(again, binned), (kind, M)
