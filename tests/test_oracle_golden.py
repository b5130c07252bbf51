"""The CPU oracle against the golden vectors captured from the reference's own modules
(tests/golden/make_golden.py).  Runs without a GPU."""
import numpy as np
import pytest
import torch

from conftest import golden, table_from_seed
from oracle import hash_encode as oh
from oracle import decoders as od
from oracle import render as orr
from oracle import lin_assign as ola


def test_g2_resolutions_fp32_quirk():
    g = golden("g2_resolutions.npz")
    for key in g.files:
        _, a, b, L = key.split("_")
        got = oh.level_resolutions(int(a), int(b), int(L))
        assert np.array_equal(np.array(got, np.float32), g[key]), key
    assert oh.level_resolutions(16, 2048, 16)[-1] == 2047.0        # Appendix E.7


def test_g1_hash_indices_bit_exact_and_feats():
    g = golden("g1_hash.npz")
    for tag in ("a", "b"):
        log2T = int(g[f"{tag}_log2T"])
        res = g[f"{tag}_res"]
        L = len(res)
        tab = torch.from_numpy(table_from_seed(int(g[f"{tag}_seed"]), (L, 2 ** log2T, 2), str(g[f"{tag}_kind"])))
        x = torch.from_numpy(g[f"{tag}_x"])
        feats, idx = oh.hash_encode(x, tab, [float(r) for r in res], log2T)
        assert np.array_equal(idx, g[f"{tag}_idx"]), "hash indices must be bit-exact"
        assert idx.min() >= 0 and idx.max() < 2 ** log2T
        assert np.array_equal(feats.numpy(), g[f"{tag}_feats"]), "fp32 features must match the reference bit for bit on CPU"


def _g3_params(g):
    p = {}
    for short, name in (("density", "decoder_density"), ("color", "decoder_color"),
                        ("semantics", "decoder_semantics"), ("inst", "decoder_inst")):
        n = int(g[f"{name}_n"])
        p[short] = ([torch.from_numpy(g[f"{name}_w{i}"]) for i in range(n)],
                    [torch.from_numpy(g[f"{name}_b{i}"]) for i in range(n)])
    return p


def test_g3_nef_forward():
    g = golden("g3_nef.npz")
    L, log2T = int(g["L"]), int(g["log2T"])
    res = [float(r) for r in oh.level_resolutions(int(g["res"][0]), int(g["res"][-1]), L)]
    coords = torch.from_numpy(g["coords"]).reshape(-1, 3)
    tabs = [torch.from_numpy(table_from_seed(int(g[k]), (L, 2 ** log2T, 2), "normal") * np.float32(0.5))
            for k in ("seed_main", "seed_delta")]
    feats, _ = oh.hash_encode(coords, tabs[0], res, log2T)
    dfeats, _ = oh.hash_encode(coords, tabs[1], res, log2T)
    assert np.array_equal(feats.numpy(), g["feats"]) and np.array_equal(dfeats.numpy(), g["delta_feats"])
    params = _g3_params(g)
    assert params["density"][0][0].shape == (64, 2 * L) and params["density"][0][1].shape == (16, 64)
    assert params["color"][0][0].shape == (64, 16 + int(g["view_embed_dim"])) and int(g["view_embed_dim"]) == 27
    assert params["inst"][0][-1].shape == (200, 64) and len(params["inst"][0]) == 3
    assert float(g["bias0"]) == 1.0
    out = od.nef_forward(feats, dfeats, torch.from_numpy(g["ray_d"]), params,
                         {"density", "rgb", "semantics", "inst_embedding"}, lod_weights=torch.ones(2 * L))
    np.testing.assert_allclose(out["density"].numpy(), g["density"].reshape(-1, 1), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(out["rgb"].numpy(), g["rgb"].reshape(-1, 3), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(out["semantics"].numpy(), g["semantics"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(out["inst_embedding"].numpy(), g["inst_embedding"], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(g["density_only"], g["density"])


def test_g4_tracer_composite_both_backgrounds():
    g = golden("g4_tracer.npz")
    N, S = int(g["N"]), int(g["S"])
    ridx, _, samples, depths, deltas, boundary = orr.raymarch_ray(
        torch.from_numpy(g["origins"]), torch.from_numpy(g["dirs"]), 0.0, 2.0, S,
        torch.from_numpy(g["jitter"]), torch.from_numpy(g["occ"]), 3)
    keep = ridx != int(g["empty_ray"])
    assert np.array_equal(ridx[keep].numpy(), g["ridx"])
    np.testing.assert_array_equal(samples[keep].numpy(), g["samples"])
    t = lambda k: torch.from_numpy(g[k])
    for bg in ("white", "black"):
        out = orr.composite(N, t("ridx"), t("boundary"), t("density"), t("deltas"), depths=t("depths"),
                            rgb=t("rgb"), semantics=t("semantics"), inst=t("inst_embedding"), bg_color=bg,
                            ray_sparcity_reg=0.01)
        for ch in ("rgb", "alpha", "depth", "semantics", "inst_embedding"):
            np.testing.assert_allclose(out[ch].numpy(), g[f"{bg}_{ch}"], rtol=1e-5, atol=1e-6, err_msg=f"{bg} {ch}")
        assert np.array_equal(out["hit"].numpy(), g[f"{bg}_hit"])
        np.testing.assert_allclose(out["ray_sparcity_loss"].numpy(), g[f"{bg}_ray_sparcity_loss"], rtol=1e-5)
    e = int(g["empty_ray"])            # Appendix E.10: empty ray keeps the background
    assert np.all(g["white_rgb"][e] == 1.0) and np.all(g["black_rgb"][e] == 0.0) and not g["white_hit"][e]
    assert g["white_alpha"][e] == 0 and g["white_depth"][e] == 0 and np.all(g["white_inst_embedding"][e] == 0)


def test_g4_voxel_travel_filter():
    g = golden("g4_tracer.npz")
    ridx, depths = torch.from_numpy(g["v_ridx"]), torch.from_numpy(g["v_depths"])
    mask = orr.voxel_travel_filter(ridx, depths, float(g["v_max_travel"]))
    assert int(mask.sum()) == int(g["v_kept"])
    k = depths.shape[1]
    deltas = torch.from_numpy(g["v_deltas"]).reshape(depths.shape)[mask].reshape(-1, 1)
    boundary = torch.from_numpy(g["v_boundary"]).reshape(depths.shape[:2])[mask].reshape(-1)
    dens = torch.from_numpy(g["v_density"])[mask]
    ridx_k = ridx[mask].repeat_interleave(k)
    out = orr.composite(16, ridx_k, boundary, dens.reshape(-1, 1), deltas, depths=depths[mask].reshape(-1, 1),
                        rgb=torch.from_numpy(g["v_rgb"])[mask].reshape(-1, 3))
    np.testing.assert_allclose(out["rgb"].numpy(), g["v_out_rgb"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(out["alpha"].numpy(), g["v_out_alpha"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(out["depth"].numpy(), g["v_out_depth"], rtol=1e-5, atol=1e-6)


def test_g5_linear_assignment_bit_exact():
    g = golden("g5_linassign.npz")
    prob, gt, stuff, pts = g["prob"], g["gt"], g["stuff"], g["points_3d"]
    for b in range(prob.shape[0]):
        sm = torch.softmax(torch.from_numpy(g["logits"][b]), -1).numpy()
        assert np.array_equal(ola.virtual_labels(sm, gt[b]), g["virt_plain"][b])
        vm = stuff[b] | (gt[b] > 0)
        assert np.array_equal(ola.virtual_labels_things(prob[b][vm], gt[b][vm]), g[f"virt_things_{b}"])
        assert np.array_equal(ola.virtual_labels_things(prob[b][vm], gt[b][vm], pts[b][vm], True),
                              g[f"virt_things_rej_{b}"])


def test_g6_sigma_sparsity():
    g = golden("g6_reg.npz")
    s = torch.from_numpy(g["sigma"])
    np.testing.assert_allclose(torch.log(1.0 + 2 * s ** 2).numpy(), g["sparsity"], rtol=1e-6)


def test_g6_segment_consistency_regularizer():
    """oracle/regularizers.py against the reference's loss/regularizers.py:5-35 (value and autograd gradient from the reference function,
    called as pc_nerf/trainer.py:525-527 calls it): every distinct id - 0 included - is a segment, all-column-0 segments are skipped,
    a 2:1 majority for column 0 forces label 0, the running total is divided by each image's segment count in turn."""
    from oracle import regularizers as oreg
    g = golden("g6_reg.npz")
    x = g["seg_prob"] + np.float32(1e-27)
    val, grad = oreg.segment_consistency_regularizer(x, g["seg_labels"], want_grad=True)
    np.testing.assert_allclose(val, g["seg_reg"], rtol=2e-6)
    assert np.array_equal(grad != 0, g["seg_reg_grad"] != 0)                       # the same (ray, column) entries carry a gradient
    np.testing.assert_allclose(grad, g["seg_reg_grad"], rtol=2e-6, atol=0)
    kinds = [best for b in range(x.shape[0]) for _, _, best in oreg.segment_labels(x[b], g["seg_labels"][b])]
    assert None in kinds and 0 in kinds and any(k not in (None, 0) for k in kinds)  # the fixture exercises all three branches
    np.testing.assert_allclose(oreg.sigma_sparsity_loss(g["sigma"]), g["sparsity"], rtol=1e-6)


def test_g7_hash_input_and_table_gradients():
    """oracle d/d xyz and d/d tables of the hash encoder vs autograd through the reference's HashGridTorch."""
    import torch
    from oracle import hash_encode as oh
    g = golden("g7_hash_grad.npz")
    res, log2T = [float(r) for r in g["res"]], int(g["log2T"])
    tab = torch.from_numpy(table_from_seed(int(g["seed"]), (len(res), 2 ** log2T, 2), str(g["kind"])))
    x, go = torch.from_numpy(g["x"]), torch.from_numpy(g["go"])
    feats, _ = oh.hash_encode(x, tab, res, log2T)
    assert np.array_equal(feats.numpy(), g["feats"])
    dx = oh.hash_encode_bwd_xyz(x, tab, go, res, log2T)
    np.testing.assert_allclose(dx.numpy(), g["dx"], rtol=1e-5, atol=1e-4)
    dt = oh.hash_encode_bwd(x, go, 2 ** log2T, res, log2T)
    np.testing.assert_allclose(dt.numpy(), g["dtables"], rtol=1e-5, atol=1e-5)


def test_permuto_position_gradient_matches_finite_differences():
    """closed-form d/d xyz of the permutohedral oracle vs central differences of its float64 restatement."""
    from oracle import permuto_encode as op
    rs = np.random.RandomState(4)
    Lv, F, cap, M = 6, 2, 512, 400
    sf = op.scale_factors(np.geomspace(1.0, 0.05, Lv))
    shifts = (rs.standard_normal(size=(Lv, 3)) * 10).astype(np.float32)
    tab = rs.standard_normal(size=(Lv, cap, F)).astype(np.float32)
    x = rs.uniform(-1, 1, size=(M, 3)).astype(np.float32)
    go = rs.standard_normal(size=(M, Lv * F)).astype(np.float32)
    f32, _, _ = op.permuto_encode(x, tab, shifts, sf)
    np.testing.assert_allclose(op.permuto_encode_f64(x, tab, shifts, sf), f32, rtol=0, atol=2e-4)
    dx = op.permuto_encode_bwd_xyz(x, tab, go, shifts, sf)
    h = 1e-6
    fd = np.zeros((M, 3))
    for a in range(3):
        e = np.zeros(3)
        e[a] = h
        fp = op.permuto_encode_f64(x.astype(np.float64) + e, tab, shifts, sf)
        fm = op.permuto_encode_f64(x.astype(np.float64) - e, tab, shifts, sf)
        fd[:, a] = ((fp - fm) * go).sum(1) / (2 * h)
    # points within h of a simplex face see two different affine pieces: drop the few outliers, demand the rest match
    err = np.abs(fd - dx).max(1) / (1.0 + np.abs(fd).max(1))
    assert np.mean(err < 1e-3) > 0.99, np.sort(err)[-10:]


def _g8_params(g):
    params = {}
    for short, name in (("density", "decoder_density"), ("color", "decoder_color"), ("semantics", "decoder_semantics"),
                        ("inst", "decoder_inst"), ("delta_density", "decoder_delta_density")):
        n = int(g[f"{name}_n"])
        params[short] = ([torch.from_numpy(g[f"{name}_w{i}"]) for i in range(n)], [torch.from_numpy(g[f"{name}_b{i}"]) for i in range(n)])
    return params


def test_g8_delta_density_variant():
    """oracle restatement of the delta-density nef + tracer vs the reference's pc_nerf/panoptic_dd_nef.py and
    tracers/panoptic_dd_packed_rf_tracer.py (golden g8)."""
    from oracle import hash_encode as oh, decoders as od, render as orr
    g = golden("g8_dd.npz")
    log2T, L = int(g["log2T"]), int(g["L"])
    res = [float(r) for r in oh.level_resolutions(int(g["res"][0]), int(g["res"][-1]), L)]
    x = torch.from_numpy(g["coords"]).reshape(-1, 3)
    tabs = [torch.from_numpy(table_from_seed(int(g[k]), (L, 2 ** log2T, 2), "normal") * np.float32(0.5)) for k in ("seed_main", "seed_delta")]
    f, _ = oh.hash_encode(x, tabs[0], res, log2T)
    df, _ = oh.hash_encode(x, tabs[1], res, log2T)
    chans = {"density", "rgb", "delta_density", "panoptic_density", "semantics", "inst_embedding"}
    out = od.nef_forward_dd(f, df, torch.from_numpy(g["ray_d"]), _g8_params(g), chans)
    for ch in chans:
        np.testing.assert_allclose(out[ch].numpy().reshape(g["nef_" + ch].shape), g["nef_" + ch], rtol=1e-5, atol=1e-6, err_msg=ch)
    N = int(g["t_N"])
    t = lambda k: torch.from_numpy(g["t_" + k])
    for bg in ("white", "black"):
        comp = orr.composite_dd(N, t("ridx"), t("boundary"), t("density"), t("panoptic_density"), t("deltas"), depths=t("depths"),
                                rgb=t("rgb"), semantics=t("semantics"), inst=t("inst_embedding"), bg_color=bg, ray_sparcity_reg=0.01)
        for ch in ("rgb", "alpha", "depth", "semantics", "inst_embedding", "ray_sparcity_loss"):
            np.testing.assert_allclose(comp[ch].numpy(), g[f"t_{bg}_{ch}"], rtol=1e-5, atol=1e-6, err_msg=f"{bg} {ch}")
        assert np.array_equal(comp["hit"].numpy(), g[f"t_{bg}_hit"])


# ------------------------------------------------------------------------------- permutohedral encoder: independent checks
# The third-party package behind grids/permuto_grid.py:57-71 is absent, so oracle/permuto_encode.py is "parity unpinned".
# These tests narrow what that leaves open: the oracle is compared with a second implementation written from the definitions
# of Adams et al. 2010 that shares no code or formula with it (oracle/permuto_adams.py), and the hash with hand-derived values.

# k = 0; for i in 0..2: k = (k + key_i) * 2531011 mod 2^32 - worked by hand (m = 2531011, m^2 mod 2^32 = 2220443785,
# m^3 mod 2^32 = 2937900635, -m mod 2^32 = 4292436285):
#   (0,0,1): ((0*m + 0)*m + 1)*m = m                      (0,1,0): (1*m)*m = m^2            (1,0,0): ((1*m)*m)*m = m^3
#   (-1,0,0): -(m^3) mod 2^32 = 4294967296 - 2937900635   (1,1,1): m^3 + m^2 + m mod 2^32
HAND_HASH = {(0, 0, 0): 0, (0, 0, 1): 2531011, (0, 1, 0): 2220443785, (1, 0, 0): 2937900635, (-1, 0, 0): 1357066661,
             (1, 1, 1): 865908135, (2, 2, -2): 1721692226, (3, -1, -1): 2295759813, (4, -8, 12): 2608358984, (-3, 1, 1): 1999207483}


def test_permuto_hash_hand_derived_vectors():
    from oracle import permuto_encode as op, permuto_adams as pa
    assert (2531011 ** 2) % 2 ** 32 == 2220443785 and (2531011 ** 3) % 2 ** 32 == 2937900635
    assert HAND_HASH[(1, 1, 1)] == (2937900635 + 2220443785 + 2531011) % 2 ** 32
    assert HAND_HASH[(-1, 0, 0)] == 2 ** 32 - 2937900635
    for cap in (2 ** 18, 2 ** 32, 1000003, 1):
        for key, k in HAND_HASH.items():
            assert pa.lattice_hash_scalar(key, cap) == k % cap
            assert int(pa.lattice_hash(np.array([key], dtype=np.int64), cap)[0]) == k % cap
    # the oracle hashes vertex r of (rem0, rank) with key_i = rem0_i + r - 4*[rank_i > 3 - r]; rank = (0,1,2,3), rem0 = 0 is the
    # canonical simplex whose vertices are (0,0,0,0), (1,1,1,-3), (2,2,-2,-2), (3,-1,-1,-1)
    rem0, rank = np.zeros((1, 4), np.int32), np.array([[0, 1, 2, 3]], np.int32)
    for cap in (2 ** 18, 1000003):
        got = op.vertex_indices(rem0, rank, cap)[0]
        want = [HAND_HASH[k] % cap for k in ((0, 0, 0), (1, 1, 1), (2, 2, -2), (3, -1, -1))]
        assert got.tolist() == want
    # centroid of that simplex: x = (0.5, 0.5, 0.5) with unit scale factors and no shift -> E = (1.5, 0.5, -0.5, -1.5), weights 1/4
    tab = np.zeros((1, 2 ** 18, 1), np.float32)
    for v, key in zip((1.0, 2.0, 4.0, 8.0), ((0, 0, 0), (1, 1, 1), (2, 2, -2), (3, -1, -1))):
        tab[0, HAND_HASH[key] % 2 ** 18, 0] = v
    x = np.full((1, 3), 0.5, np.float32)
    f, idx, b = op.permuto_encode(x, tab, np.zeros((1, 3), np.float32), np.ones((1, 3), np.float32))
    assert f[0, 0] == 3.75 and np.array_equal(b[0, 0], np.full(4, 0.25, np.float32))


def _permuto_setup(L=24, cap=1 << 18, F=2, seed=0):
    from oracle import permuto_encode as op
    rs = np.random.RandomState(seed)
    sf = op.scale_factors(np.geomspace(1.0, 1e-4, L))                      # grids/permuto_grid.py:53 at best.yaml sizes
    shifts = (rs.standard_normal((L, 3)) * 10).astype(np.float32)
    tab = rs.standard_normal((L, cap, F)).astype(np.float32)
    return rs, sf, shifts, tab


def _check_levels(x, levels, sf, shifts, tab, min_same):
    from oracle import permuto_encode as op, permuto_adams as pa
    F = tab.shape[2]
    sub = lambda a: np.ascontiguousarray(a[levels])
    f32, i32, b32 = op.permuto_encode(x, sub(tab), sub(shifts), sub(sf))
    f64, i64, b64 = pa.encode(x, sub(tab), sub(shifts), sub(sf))
    for j, l in enumerate(levels):
        _, _, E = pa.enclosing_simplex(x[:1000], shifts[l], sf[l])
        ulp = float(np.spacing(np.float32(np.abs(E).max())))             # fp32 resolution of the elevated coordinates at this level
        same = (i32[j] == i64[j]).all(1)
        assert same.mean() >= min_same(l), (l, same.mean())
        # where both pick the same simplex the weights agree to the fp32 rounding of E; everywhere the FEATURES agree to it,
        # because the interpolant is continuous across faces (a different simplex near a face carries a ~0 weight there)
        assert np.abs(b32[j][same] - b64[j][same]).max() <= 4 * ulp + 2e-6, (l, ulp)
        err = np.abs(f32[:, j * F:(j + 1) * F] - f64[:, j * F:(j + 1) * F]).max()
        assert err <= (8 * ulp + 1e-5) * np.abs(tab[l]).max(), (l, err, ulp)
        assert (b64[j] >= -1e-9).all() and np.abs(b64[j].sum(1) - 1).max() < 1e-9


def test_permuto_oracle_matches_independent_adams_statement():
    rs, sf, shifts, tab = _permuto_setup()
    min_same = lambda l: 0.9999 if l <= 8 else (0.998 if l <= 18 else 0.98)
    x = rs.uniform(-1, 1, size=(1_000_000, 3)).astype(np.float32)
    _check_levels(x, [0, 9, 17], sf, shifts, tab, min_same)                 # 1e6 points on a coarse, a middle and a fine level
    _check_levels(x[:40000], list(range(24)), sf, shifts, tab, min_same)    # every level
    from oracle import permuto_encode as op
    _check_levels(op.half_round(x[:200000]), [3, 12, 23], sf, shifts, tab, min_same)   # fp16-rounded coordinates (the training default)


def test_permuto_points_on_simplex_faces_edges_and_vertices():
    """Points constructed ON faces / edges / vertices of lattice simplices (one, two or three barycentric weights zero), where
    the choice of simplex is ambiguous: both implementations must still produce the same (continuous) features."""
    from oracle import permuto_encode as op, permuto_adams as pa
    rs, sf, shifts, tab = _permuto_setup(L=12, cap=1 << 16)
    sf = op.scale_factors(np.geomspace(1.0, 1e-2, 12))
    shifts = (rs.standard_normal((12, 3)) * 2).astype(np.float32)
    for l in (0, 5, 11):
        seed_pts = rs.uniform(-1, 1, size=(30000, 3))
        verts, _, _ = pa.enclosing_simplex(seed_pts, shifts[l], sf[l])
        w = rs.dirichlet(np.ones(4), size=len(seed_pts))
        n_zero = rs.randint(1, 4, size=len(seed_pts))                       # 1: face, 2: edge, 3: vertex
        order = np.argsort(rs.rand(len(seed_pts), 4), axis=1)
        for i in range(len(seed_pts)):
            w[i, order[i, :n_zero[i]]] = 0.0
        w /= w.sum(1, keepdims=True)
        E = (w[:, :, None] * verts).sum(1)
        x = pa.unelevate(E, shifts[l], sf[l]).astype(np.float32)
        keep = (np.abs(x) <= 1.5).all(1)
        x = x[keep]
        f32, _, b32 = op.permuto_encode(x, tab[l:l + 1], shifts[l:l + 1], sf[l:l + 1])
        f64, _, b64 = pa.encode(x, tab[l:l + 1], shifts[l:l + 1], sf[l:l + 1])
        _, _, Ef = pa.enclosing_simplex(x[:100], shifts[l], sf[l])
        ulp = float(np.spacing(np.float32(np.abs(Ef).max())))
        assert np.abs(f32 - f64).max() <= (8 * ulp + 1e-5) * np.abs(tab[l]).max(), l
        assert (b32 >= -4 * ulp - 1e-6).all() and np.abs(b32.sum(-1) - 1).max() < 1e-5


def _g9_params(g, tag, requires_grad=False):
    wt = tag if tag in ("dd", "pos") else "dd"
    p = {}
    for short, name in (("density", "decoder_density"), ("color", "decoder_color"), ("semantics", "decoder_semantics"), ("inst", "decoder_inst")):
        n = int(g[f"{wt}_{name}_n"])
        p[short] = ([torch.from_numpy(g[f"{wt}_{name}_w{i}"]).clone().requires_grad_(requires_grad) for i in range(n)],
                    [torch.from_numpy(g[f"{wt}_{name}_b{i}"]).clone().requires_grad_(requires_grad) for i in range(n)])
    return p


def g9_oracle(g, tag):
    """-> (channels dict, table gradient, params) of oracle.decoders.nef_forward_base + oracle.hash_encode under torch autograd for one
    configuration of golden g9 (the reference's BASE field pc_nerf/panoptic_nef.py::PanopticNeF)."""
    L, log2T = int(g["L"]), int(g["log2T"])
    res = [float(r) for r in oh.level_resolutions(int(g["res"][0]), int(g["res"][-1]), L)]
    coords = torch.from_numpy(g["coords"]).reshape(-1, 3)
    tab = torch.from_numpy(table_from_seed(int(g["seed_main"]), (L, 2 ** log2T, 2), "normal") * np.float32(0.5)).requires_grad_(True)
    feats, _ = oh.hash_encode(coords, tab, res, log2T)
    params = _g9_params(g, tag, requires_grad=True)
    sd, idt, direct = dict(dd=(True, True, False), ll=(False, False, False), ld=(False, True, False), pos=(True, True, True))[tag]
    chans = {"density", "rgb", "inst_embedding"} | (set() if direct else {"semantics"})
    out = od.nef_forward_base(feats, torch.from_numpy(g["ray_d"]), params, chans, sem_detach=sd, inst_detach=idt, inst_direct_pos=direct,
                              coords=coords)
    loss = sum((out[c] * torch.from_numpy(g["G_" + c]).reshape(out[c].shape)).sum() for c in chans)
    loss.backward()
    return {c: out[c] for c in chans}, tab.grad, params


@pytest.mark.parametrize("tag", ["dd", "ll", "ld", "pos"])
def test_g9_base_nef_channels_and_gradients(tag):
    """oracle.decoders.nef_forward_base against the reference's PanopticNeF (golden g9): channels, and the gradients of a fixed linear
    functional of them - with sem_detach / inst_detach off the panoptic term reaches the grid tables (the table gradient changes), with
    both on it does not."""
    g = golden("g9_base_nef.npz")
    assert str(g[f"{tag}_nef_type"]) == "panoptic_nef"
    out, dtab, params = g9_oracle(g, tag)
    for c, v in out.items():
        np.testing.assert_allclose(v.detach().numpy().reshape(g[f"{tag}_{c}"].shape), g[f"{tag}_{c}"], rtol=1e-5, atol=1e-7)
    want = g[f"{tag}_dtables"]
    np.testing.assert_allclose(dtab.numpy(), want, rtol=1e-4, atol=1e-5 * float(np.abs(want).max()))
    for short, name in (("density", "decoder_density"), ("color", "decoder_color"), ("semantics", "decoder_semantics"), ("inst", "decoder_inst")):
        for i, (W, b) in enumerate(zip(*params[short])):
            if f"{tag}_{name}_dw{i}" in g.files:
                wg = g[f"{tag}_{name}_dw{i}"]
                np.testing.assert_allclose(W.grad.numpy(), wg, rtol=1e-4, atol=1e-5 * float(np.abs(wg).max()))
                np.testing.assert_allclose(b.grad.numpy(), g[f"{tag}_{name}_db{i}"], rtol=1e-4, atol=1e-5 * float(np.abs(wg).max()))
    if tag != "dd":
        rel = np.abs(want - g["dd_dtables"]).max() / np.abs(g["dd_dtables"]).max()
        assert (rel > 1e-2) == (tag in ("ll", "ld")), rel          # 'pos': the instance head never touches the grid


def test_lsap_restatement_equals_scipy():
    """oracle.lin_assign.lsap_jv - the sequential restatement of SciPy's rectangular LSAP solver that the device kernel pag_assign_solve follows - against
    scipy.optimize.linear_sum_assignment itself (the solver loss/lin_assignment_things.py:45 calls; SciPy %s here): identical columns on random, tie-heavy
    integer, constant, fp32-valued and outlier-masked (10000) matrices with rows <= columns, and on the golden batch's own cost matrices.""" % __import__("scipy").__version__
    from scipy.optimize import linear_sum_assignment
    from oracle.lin_assign import lsap_jv
    rs = np.random.RandomState(0)
    for trial in range(1200):
        nr = rs.randint(1, 13)
        nc = rs.randint(nr, 26)
        kind = trial % 6
        if kind == 0:
            c = rs.randn(nr, nc)
        elif kind == 1:
            c = rs.randint(0, 3, (nr, nc)).astype(np.float64)              # many ties
        elif kind == 2:
            c = -rs.rand(nr, nc).astype(np.float32).astype(np.float64)     # what the loss produces: -mean probability, fp32 widened
        elif kind == 3:
            c = np.zeros((nr, nc))
        elif kind == 4:
            c = -rs.rand(nr, nc).astype(np.float32).astype(np.float64)
            c[rs.rand(nr, nc) < 0.6] = 10000                               # outlier-rejection mask
        else:
            c = np.round(rs.randn(nr, nc) * 2) / 2
        rows, cols = linear_sum_assignment(c)
        assert np.array_equal(rows, np.arange(nr)) and np.array_equal(cols, lsap_jv(c)), (trial, kind)
    big = -rs.rand(60, 199).astype(np.float32).astype(np.float64)
    assert np.array_equal(linear_sum_assignment(big)[1], lsap_jv(big))


def test_lsap_restatement_equals_the_committed_scipy_answers():
    """g10_lsap.npz: 48 cost matrices with the columns scipy.optimize.linear_sum_assignment (SciPy 1.15.3, tests/golden/make_golden.py::g10) assigned - the
    definition of record for the Hungarian step should a later image ship a SciPy with another tie rule.  oracle.lin_assign.lsap_jv reproduces every one."""
    from oracle.lin_assign import lsap_jv
    g = golden("g10_lsap.npz")
    n = 0
    while "cost_%d" % n in g:
        assert np.array_equal(lsap_jv(g["cost_%d" % n].astype(np.float64)), g["cols_%d" % n]), n
        n += 1
    assert n == 48
