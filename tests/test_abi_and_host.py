"""CPU-side checks: the C-ABI library loads and exports every symbol include/pagnerf_hip.h
declares (no compute without a GPU), argument validation, and host-side logic."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import REPO


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build()
    from pagnerf_amd import _lib
    return _lib.load()


def test_every_declared_symbol_is_exported(lib):
    from pagnerf_amd import _lib
    hdr = open(os.path.join(REPO, "include", "pagnerf_hip.h")).read()
    declared = set(re.findall(r"\b(pag_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.pag_abi_version() == _lib.ABI_VERSION == 14


def test_assignment_entry_points_validate_without_gpu(lib):
    """pag_assign_cost / pag_assign_nll_fwd / _bwd (ABI 10): empty batches are no-ops, bad sizes and NULL buffers are rejected before any launch."""
    assert lib.pag_assign_nll_fwd(None, 1, 0, 0, 200, 200, None, None, None, None, None, 199, 1, None, None, None, None, None) == 0
    assert lib.pag_assign_nll_bwd(None, 0, 5, 0, 200, 200, None, None, None, None, None, None) == 0
    tail = (None, 0.0, 0.0, 0, None, None, None, None)              # no outlier rejection: points, id_slope, id_x_limit, id_margin, psums_ws, pcounts_ws, id_lo_hi; stream
    assert lib.pag_assign_cost(None, 0, 5, 0, 200, 200, 1, None, 199, None, None, None, None, None, *tail) == 0               # no image: nothing to do
    assert lib.pag_assign_cost(None, 1, 5, 0, 200, 200, 1, None, 199, None, None, None, None, None, *tail) == -1              # NULL buffers
    assert b"NULL" in lib.pag_last_error_string()
    buf = (ctypes.c_float * 8)()
    assert lib.pag_assign_cost(buf, 1, 5, 0, 100, 200, 1, buf, 199, buf, buf, buf, buf, buf, *tail) == -1                     # row_stride < n_cols
    assert lib.pag_assign_cost(buf, 1, 5, 0, 200, 200, 1, buf, 2000, buf, buf, buf, buf, buf, *tail) == -1                    # max_rows > 1024
    assert b"max_rows" in lib.pag_last_error_string()
    assert lib.pag_assign_nll_fwd(buf, 1, 5, 0, 100, 200, buf, None, buf, buf, buf, 199, 1, buf, buf, buf, buf, None) == -1
    assert lib.pag_assign_nll_bwd(buf, 1, 5, 0, 200, 200, buf, buf, buf, None, buf, None) == -1                              # NULL grad


def test_pose_entry_points_validate_without_gpu(lib):
    """pag_pose_rays_fwd / _bwd, pag_view_embed_bwd (ABI 11): empty batches are no-ops, bad sizes and NULL buffers are rejected before any launch."""
    buf = (ctypes.c_float * 16)()
    assert lib.pag_pose_rays_fwd(None, 1, None, 1, None, None, 0, None, None, None) == 0                       # no ray: nothing to do
    assert lib.pag_pose_rays_fwd(buf, 1, None, 1, buf, buf, 4, buf, buf, None) == -1                          # NULL camera index
    assert b"NULL" in lib.pag_last_error_string()
    assert lib.pag_pose_rays_fwd(buf, 0, buf, 1, buf, buf, 4, buf, buf, None) == -1                           # no camera
    assert lib.pag_pose_rays_fwd(buf, 1, buf, 0, buf, buf, 4, buf, buf, None) == -1                           # rays_per_entry < 1
    assert b"rays_per_entry" in lib.pag_last_error_string()
    assert lib.pag_pose_rays_fwd(buf, 1, buf, 1, buf, buf, 4, None, buf, None) == -1                          # NULL output
    assert lib.pag_pose_rays_bwd(buf, 1, buf, 1, buf, buf, 4, buf, buf, None, buf, 1 << 20, None) == -1        # NULL d_params
    assert lib.pag_pose_rays_bwd(buf, 1, buf, 1, buf, buf, 4, buf, buf, buf, buf, 8, None) == -1               # workspace too small
    assert b"workspace" in lib.pag_last_error_string() and lib.pag_pose_rays_bwd_workspace_bytes(6) == 6 * 32 * 12 * 4
    assert lib.pag_pose_points(None, 1, None, 1, None, None, None, 0, None, None) == 0                         # no ray
    assert lib.pag_pose_points(buf, 1, buf, 1, buf, buf, None, 4, buf, None) == -1 and b"depth" in lib.pag_last_error_string()
    # segment regulariser (ABI 12): sizes, NULL buffers and a short workspace are refused before any launch
    need = lib.pag_segment_reg_workspace_bytes(6, 4096)
    assert need >= 2 * 6 * 4096 * 4 + 3 * 6 * 2048 * 4
    assert lib.pag_segment_reg_fwd(buf, 0, 8, 16, 2, 2, 0.0, buf, buf, need, buf, None) == -1                  # B < 1
    assert lib.pag_segment_reg_fwd(buf, 1, 8, 16, 1, 2, 0.0, buf, buf, need, buf, None) == -1                  # row_stride < n_cols
    assert lib.pag_segment_reg_fwd(buf, 1, 8, 16, 2, 2, 0.0, None, buf, need, buf, None) == -1 and b"NULL" in lib.pag_last_error_string()
    assert lib.pag_segment_reg_fwd(buf, 1, 8, 16, 2, 2, 0.0, buf, buf, 64, buf, None) == -1 and b"workspace" in lib.pag_last_error_string()
    assert lib.pag_segment_reg_bwd(buf, 1, 8, 16, 2, 2, 0.0, buf, 64, buf, buf, None) == -1 and b"workspace" in lib.pag_last_error_string()
    # the device Hungarian step (ABI 13): shapes beyond one wave's 256 x 256 and NULL buffers are refused before any launch; no image = no-op
    assert lib.pag_assign_solve(None, 0, 199, 199, None, None, None, None, None) == 0
    assert lib.pag_assign_solve(buf, 1, 257, 199, buf, None, buf, buf, None) == -1 and b"max_rows" in lib.pag_last_error_string()
    assert lib.pag_assign_solve(buf, 1, 199, 300, buf, None, buf, buf, None) == -1
    assert lib.pag_assign_solve(buf, 1, 199, 199, buf, None, None, buf, None) == -1 and b"NULL" in lib.pag_last_error_string()
    # the touched-rows exchange's passes (ABI 14): sizes and NULL buffers are refused before any launch
    assert lib.pag_sparse_rows_mask(buf, 0, 64, 2, buf, None) == -1 and lib.pag_sparse_rows_mask(buf, 4, 0, 2, buf, None) == -1
    assert lib.pag_sparse_rows_mask(buf, 4, 64, 65, buf, None) == -1 and b"F 65" in lib.pag_last_error_string()
    assert lib.pag_sparse_rows_mask(None, 4, 64, 2, buf, None) == -1 and b"NULL" in lib.pag_last_error_string()
    assert lib.pag_sparse_rows_plan(buf, 4, 64, None, buf, buf, None) == -1
    assert lib.pag_sparse_rows_pack(buf, 4, 64, 2, buf, buf, buf, None, buf, None) == -1 and lib.pag_sparse_rows_unpack(None, 4, 64, 2, buf, buf, buf, buf, buf, None) == -1
    assert lib.pag_view_embed_bwd(None, 0, 4, 32, None, None, None) == 0
    assert lib.pag_view_embed_bwd(buf, 2, 4, 16, buf, buf, None) == -1                                        # width < 3 + 6 n_freq
    assert lib.pag_view_embed_bwd(buf, 2, 4, 32, None, buf, None) == -1


def test_regular_library_carries_no_instrumentation(lib):
    """The timing builds (-DPAG_REDUCE_TIMING / PAG_BIN_TIMING / PAG_BLOCK_TIMING: scripts/reduce_phases.py, bin_phases.py, block_timeline.py)
    export pag_debug_* readers; the library the product loads must not - nothing in it stamps clocks or writes debug tables."""
    for name in ("pag_debug_reduce_times", "pag_debug_reduce_occupancy", "pag_debug_bin_times", "pag_debug_block_times_encode", "pag_debug_block_times_mlp"):
        assert not hasattr(lib, name), name
    assert not os.environ.get("PAG_LIB_VARIANT"), "tests must run against the regular library"


def test_argument_validation_without_gpu(lib):
    from pagnerf_amd import _lib as L
    f = (ctypes.c_float * 4)(16, 32, 64, 128)
    # M == 0 is a no-op for every entry point (empty packs / empty batches, permuto_grid.py:68-69)
    assert lib.pag_hash_encode_fwd(None, 0, None, L.F32, 4, 2, 10, f, None, None, L.F32, 8, 1, 0, 0, None) == 0
    assert lib.pag_composite_fwd(None, None, 0, None, None, None, None, 1, None, None, None, None, None, 0, None) == 0
    assert lib.pag_raymarch_count(None, None, 0, 8, None, None, 0.0, 2.0, None, 7, None, None) == 0
    # bad arguments are rejected before any launch
    assert lib.pag_hash_encode_fwd(None, 5, None, L.F32, 4, 2, 10, f, None, None, L.F32, 8, 1, 0, 0, None) == -1
    assert b"xyz" in lib.pag_last_error_string()
    assert lib.pag_hash_encode_fwd(None, 0, None, L.F32, 99, 2, 10, f, None, None, L.F32, 8, 1, 0, 0, None) == -1
    assert lib.pag_hash_encode_fwd(None, 0, None, L.F32, 4, 3, 10, f, None, None, L.F32, 8, 1, 0, 0, None) == -1
    a = L.MlpFwdArgs()
    a.n_layers, a.k1, a.in_dim, a.out_dim = 5, 48, 48, 16
    assert lib.pag_mlp_fwd(ctypes.byref(a), 0, None) == -1 and b"n_layers" in lib.pag_last_error_string()
    a.n_layers, a.out_dim = 2, 500
    assert lib.pag_mlp_fwd(ctypes.byref(a), 0, None) == -1 and b"out_dim" in lib.pag_last_error_string()
    # the single-launch helpers around the path
    assert lib.pag_view_embed(None, 0, 4, 32, None, None) == 0                    # R == 0: no-op
    assert lib.pag_view_embed(None, 8, 4, 16, None, None) == -1 and b"width" in lib.pag_last_error_string()
    assert lib.pag_pack_offsets(None, -1, None, None, None) == -1
    assert lib.pag_mlp_wgrad_batch(None, 0, 8, None) == -1 and b"n_layers" in lib.pag_last_error_string()
    layers = (L.WgradLayer * 1)()
    layers[0].n_out, layers[0].dz_cols = 300, 300
    assert lib.pag_mlp_wgrad_batch(layers, 1, 8, None) == -1 and b"n_out" in lib.pag_last_error_string()
    assert lib.pag_render_loss_workspace_bytes() >= 64
    t = [None, 0, None, None, 0.0, 1.0, 0]
    one = ctypes.c_void_p(16)        # a non-NULL pointer value; rejected before it is ever dereferenced
    assert lib.pag_render_loss_fwd(one, None, 4, 1.0, *t, *t, 1e-27, one, one, None) == -1 and b"rgb_gt" in lib.pag_last_error_string()
    assert lib.pag_render_loss_fwd(None, None, 4, 1.0, one, 0, None, None, 1.0, 1.0, 0, *t, 1e-27, one, one, None) == -1
    assert lib.pag_render_loss_bwd(None, None, None, None, 4, 1.0, *t, *t, 1e-27, None, None, None, None) == -1
    # round-3 entry points
    assert lib.pag_ray_sample_grad(None, None, 0, None, None, None, None) == 0                        # no packs: no-op
    assert lib.pag_ray_sample_grad(None, None, -1, None, None, None, None) == -1
    assert lib.pag_ray_sample_grad(None, None, 4, None, None, None, None) == -1 and b"NULL" in lib.pag_last_error_string()
    assert lib.pag_pad_packed(None, -1, 0, 1, None, None, None, None, None, None, None, None, None, None) == -1
    assert lib.pag_adam_step(2, None, None, None, None, None, 1e-3, 0.9, 0.999, 1e-15, 0.0, 1, None) == -1                 # NULL lists
    assert lib.pag_adam_step(0, None, None, None, None, None, 1e-3, 0.9, 0.999, 1e-15, 0.0, 0, None) == -1                 # step counts from 1
    assert lib.pag_adam_step(0, None, None, None, None, None, 1e-3, 1.0, 0.999, 1e-15, 0.0, 1, None) == -1                 # beta1 < 1
    assert lib.pag_adam_step(0, None, None, None, None, None, 1e-3, 0.9, 0.999, 1e-15, 0.0, 1, None) == 0                  # nothing to do


def test_product_path_refuses_cpu_tensors():
    from pagnerf_amd import ops
    spec = ops.hash_spec([16.0, 32.0], 8, 2)
    with pytest.raises(RuntimeError, match="GPU"):
        ops.encode(torch.zeros(4, 3), torch.zeros(2, 256, 2), spec)
    with pytest.raises(RuntimeError, match="GPU"):
        ops.fused_mlp(torch.zeros(4, 48), [torch.zeros(64, 48), torch.zeros(16, 64)], [torch.zeros(64), torch.zeros(16)])


def test_package_does_not_import_oracle():
    import subprocess, sys
    code = "import sys; sys.path.insert(0, %r); import pagnerf_amd; assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules), 'oracle imported'" % REPO
    subprocess.check_call([sys.executable, "-c", code])
    for root, _, files in os.walk(os.path.join(REPO, "pagnerf_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f


def test_hash_level_resolutions_match_reference_quirk():
    from pagnerf_amd import HashGridHIP
    from conftest import golden
    g = golden("g2_resolutions.npz")
    for key in g.files:
        _, a, b, Lv = key.split("_")
        assert np.array_equal(np.array(HashGridHIP.level_resolutions(int(a), int(b), int(Lv)), np.float32), g[key])


def test_occupancy_bitfield_roundtrip_and_state_dict():
    from pagnerf_amd.grids import OccupancyBLAS, PermutoGridHIP
    blas = OccupancyBLAS(4)
    assert blas.occupancy_mask().all() and blas.dense_points.shape == (4096, 3)
    m = torch.rand(4096) > 0.5
    blas.blas_init(m)
    assert torch.equal(blas.occupancy_mask(), m)
    # linear order of dense_points == bit order (x slowest)
    p = blas.dense_points.long()
    assert torch.equal((p[:, 0] * 16 + p[:, 1]) * 16 + p[:, 2], torch.arange(4096))
    g = PermutoGridHIP(2, capacity_log_2=8, num_lods=4, finest_scale=0.01, blas_level=3)
    g.init_from_scales()
    sd = g.state_dict()
    assert set(sd) == {"tables", "blas_bits", "random_shift_per_level"} and sd["tables"].shape == (4, 256, 2)
    import copy
    g2 = copy.deepcopy(g)
    g2.set_capacity(6)
    g2.init_from_scales()
    assert g2.tables.shape == (4, 64, 2) and g.tables.shape == (4, 256, 2)


def test_nef_and_tracer_api_surface():
    import pagnerf_amd
    nef = pagnerf_amd.PanopticDeltaNeF(grid_type="PermutoGrid", num_lods=24, feature_dim=2, num_classes=6, num_instances=200,
                                       inst_num_layers=2, sem_num_layers=1, sem_softmax=True, inst_softmax=True,
                                       panoptic_features_type="delta", capacity_log_2=8, delta_capacity_log_2=6, blas_level=3,
                                       some_unrelated_cli_flag=1)
    nef.grid.init_from_scales()
    nef.delta_grid.init_from_scales()
    assert nef.get_supported_channels() == {"density", "rgb", "semantics", "inst_embedding"}
    assert nef.grid.tables.shape == (24, 256, 2) and nef.delta_grid.tables.shape == (24, 64, 2)
    shapes = {n: tuple(p.shape) for n, p in nef.named_parameters()}
    assert shapes["decoder_density.layers.0.weight"] == (64, 48) and shapes["decoder_density.lout.weight"] == (16, 64)
    assert shapes["decoder_color.layers.0.weight"] == (64, 43) and shapes["decoder_color.lout.weight"] == (3, 64)
    assert shapes["decoder_semantics.lout.weight"] == (6, 64) and shapes["decoder_inst.lout.weight"] == (200, 64)
    assert "decoder_inst.layers.1.weight" in shapes and float(nef.decoder_density.lout.bias[0]) == 1.0
    assert sum(int(np.prod(s)) for n, s in shapes.items() if "decoder" in n) == 35169
    assert any("grid" in n for n in shapes) and any("delta_grid" in n for n in shapes)          # trainer.py:250-255 LR groups
    tr = pagnerf_amd.PanopticPackedRFTracer(raymarch_type="ray", num_steps=512, bg_color="white", ray_max_travel=6.0)
    assert tr.get_supported_channels() == {"depth", "hit", "rgb", "alpha", "semantics", "inst_embedding"}
    assert tr.get_required_nef_channels() == {"rgb", "density"}
    with pytest.raises(Exception):
        nef(channels={"nonexistent"}, coords=torch.zeros(1, 1, 3))
    rb = pagnerf_amd.RenderBuffer(rgb=torch.zeros(4, 3), alpha=torch.zeros(4, 1))
    rb += pagnerf_amd.RenderBuffer(rgb=torch.ones(2, 3), alpha=torch.ones(2, 1))
    assert rb.rgb.shape == (6, 3) and rb.reshape(2, 3, -1).rgb.shape == (2, 3, 3)
    rays = pagnerf_amd.Rays(torch.zeros(10, 3), torch.ones(10, 3), 0.0, 2.0)
    assert [len(r) for r in rays.split(4)] == [4, 4, 2] and rays[2:5].origins.shape == (3, 3)


def test_sample_count_mailbox_polling_host_logic():
    """ops._poll_count: returns the value once it is non-negative, -1 after the timeout (the caller then falls back to the
    synchronous read-back) - exercised on a plain numpy mailbox, no GPU."""
    import threading
    import time
    import numpy as np
    from pagnerf_amd import ops
    box = (None, np.array([-1], dtype=np.int64))
    t0 = time.perf_counter()
    assert ops._poll_count(box, timeout_s=0.05) == -1
    assert 0.04 < time.perf_counter() - t0 < 2.0
    threading.Timer(0.02, lambda: box[1].__setitem__(0, 12345)).start()
    assert ops._poll_count(box, timeout_s=5.0) == 12345
    box[1][0] = 0
    assert ops._poll_count(box) == 0                     # zero samples is a valid count


def test_state_dict_reload_rebuilds_host_side_derived_state():
    """after load_state_dict the kernels must hash with the LOADED per-level shifts and march with the LOADED
    occupancy - both live in host-side copies (`_spec`, `_all_occupied`) derived from the buffers."""
    from pagnerf_amd.grids import PermutoGridHIP
    torch.manual_seed(1)
    g = PermutoGridHIP(2, capacity_log_2=8, num_lods=4, finest_scale=0.01, blas_level=3)
    g.init_from_scales()
    g.blas_init(torch.rand(512) > 0.5)
    assert not g._all_occupied
    torch.manual_seed(2)
    h = PermutoGridHIP(2, capacity_log_2=8, num_lods=4, finest_scale=0.01, blas_level=3)
    h.init_from_scales()
    assert h._all_occupied and list(h._spec.shift) != list(g._spec.shift)
    res = h.load_state_dict(g.state_dict())
    assert not res.missing_keys and not res.unexpected_keys
    assert list(h._spec.shift) == list(g._spec.shift) == g.random_shift_per_level.reshape(-1).tolist()
    assert h._all_occupied is False and torch.equal(h.occupancy_mask(), g.occupancy_mask())
    assert h._spec.flags == g._spec.flags == 1          # half_coords is the default (grids/permuto_grid.py:65)
    # through a parent module too (pipeline.load_state_dict)
    import pagnerf_amd
    mk = lambda seed: (torch.manual_seed(seed), pagnerf_amd.PanopticDeltaNeF(
        grid_type="PermutoGrid", num_lods=4, feature_dim=2, num_classes=6, num_instances=8, inst_num_layers=2, sem_num_layers=1,
        capacity_log_2=6, delta_capacity_log_2=6, blas_level=3, panoptic_features_type="delta"))[1]
    a, b = mk(3), mk(4)
    for n in (a, b):
        n.grid.init_from_scales()
        n.delta_grid.init_from_scales()
    a.grid.blas_init(torch.rand(512) > 0.3)
    b.load_state_dict(a.state_dict())
    assert list(b.grid._spec.shift) == list(a.grid._spec.shift) and list(b.delta_grid._spec.shift) == list(a.delta_grid._spec.shift)
    assert b.grid._all_occupied is False and b.delta_grid._all_occupied is True


def test_pipeline_pickles_and_torch_saves(tmp_path):
    """`torch.save(pipeline)` is the reference's default checkpoint format (config_parser.py:753-756)."""
    import copy
    import io
    import pickle
    import pagnerf_amd
    nef = pagnerf_amd.PanopticDeltaNeF(grid_type="PermutoGrid", num_lods=4, feature_dim=2, num_classes=6, num_instances=8,
                                       inst_num_layers=2, sem_num_layers=1, capacity_log_2=6, delta_capacity_log_2=5, blas_level=3,
                                       panoptic_features_type="delta")
    nef.grid.init_from_scales()
    nef.delta_grid.init_from_scales()
    pipe = pagnerf_amd.Pipeline(nef, pagnerf_amd.PanopticPackedRFTracer(raymarch_type="ray", num_steps=16))
    blob = pickle.dumps(pipe)
    back = pickle.loads(blob)
    assert list(back.nef.grid._spec.shift) == list(nef.grid._spec.shift) and back.nef.grid._spec.capacity == 64
    assert back.nef.delta_grid._spec.capacity == 32 and back.nef.grid._spec.flags == nef.grid._spec.flags
    path = tmp_path / "model.pth"
    torch.save(pipe, path)
    again = torch.load(path, weights_only=False)
    assert torch.equal(again.nef.grid.tables, nef.grid.tables) and again.tracer.num_steps == 16
    hs = pagnerf_amd.HashGridHIP(2, codebook_bitwidth=6, blas_level=3)
    hs.init_from_resolutions([16, 32, 64])
    hs2 = copy.deepcopy(hs)
    assert list(hs2._spec.res) == list(hs._spec.res) and pickle.loads(pickle.dumps(hs))._spec.log2_T == 6


def test_render_buffer_channels_are_real_attributes():
    """the reference trainer finds channels with dir() / vars() (trainer.py:438, :488)."""
    import pagnerf_amd
    rb = pagnerf_amd.RenderBuffer(rgb=torch.zeros(4, 3), inst_embedding=torch.ones(4, 5), ray_sparcity_loss=torch.tensor(0.5))
    assert "ray_sparcity_loss" in dir(rb) and vars(rb)["inst_embedding"].shape == (4, 5)
    assert rb.channels == {"rgb", "inst_embedding", "ray_sparcity_loss"}
    rb.depth = torch.zeros(4, 1)
    assert "depth" in vars(rb) and rb.cpu().depth.shape == (4, 1)
    rb += pagnerf_amd.RenderBuffer(rgb=torch.ones(2, 3), inst_embedding=torch.ones(2, 5), depth=torch.ones(2, 1))
    assert rb.rgb.shape == (6, 3) and float(rb.ray_sparcity_loss) == 0.5
    with pytest.raises(AttributeError):
        rb.nonexistent


def test_mailbox_pool_never_lends_one_word_twice():
    """concurrent marches (threads / streams) each get their own pinned word."""
    import threading
    from pagnerf_amd import ops
    if not torch.cuda.is_available():
        # pin_memory needs the HIP runtime: exercise the pool logic with plain words
        real = torch.Tensor.pin_memory
        torch.Tensor.pin_memory = lambda self, *a, **k: self
    try:
        held, lock = [], threading.Lock()

        def worker():
            for _ in range(50):
                m = ops._count_mailbox()
                with lock:
                    assert all(m[0] is not h[0] for h in held)
                    held.append(m)
                m[1][0] = 7
                with lock:
                    held.remove(m)
                ops._release_mailbox(m)
        ts = [threading.Thread(target=worker) for _ in range(4)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert len(ops._MAILBOX_FREE) <= 16
    finally:
        if not torch.cuda.is_available():
            torch.Tensor.pin_memory = real


def test_bench_launches_its_own_ranks_dry_run():
    """`python bench.py --gpus 2` (no WORLD_SIZE) must start two ranks itself, prove the process group saw
    both, run the shard collectives over it and relay rank 0's line; a size mismatch must fail instead of reporting."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import tempfile
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    detail_path = os.path.join(tempfile.mkdtemp(), "bench_detail.json")
    env["PAG_BENCH_DETAIL"] = detail_path          # the full record goes to a side file; stdout carries the compact line only
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and r.stdout.strip().splitlines()[-1] == lines[0]          # the result line is the LAST stdout line
    assert len(lines[0]) < 4096
    small = json.loads(lines[0])
    assert small["n_gpus"] == 2 and small["rccl_ranks_seen"] == 2 and small["dry_run"] is True and small["steps"] == 3 and small["warmup"] == 1
    assert "roofline" in small and "cpu_baseline" in small and small["detail"] == detail_path
    assert {v["grad_sync"] for v in small["weak_regimes"].values()} == {"fp32", "bf16"}
    d = json.load(open(detail_path))
    assert d["n_gpus"] == 2 and d["rccl_ranks_seen"] == 2 and d["dry_run"] is True and d["steps"] == 3 and d["warmup"] == 1
    # the weak-scaling regime lines: GradSync(comm_dtype="auto") keeps the fp32 all-reduce for the long (dense) step and switches to the bf16 direct
    # reduce for the short post-prune steps (step < 4 x the predicted exposed exchange), every rank at the same step
    weak = {w["name"]: w["grad_sync"] for w in d["weak_regimes"]}
    assert weak["weak_dense_all_channels"]["comm_dtype"] == "fp32" and weak["weak_post_prune_rgb"]["comm_dtype"] == "bf16"
    assert weak["weak_post_prune_all_channels"]["comm_dtype"] == "bf16" and all(abs(w["predicted_fp32_exchange_ms"] - 15.0) < 1e-6 for w in weak.values())
    # ... and the touched-rows exchange per regime: the dense regime moves its whole table, the post-prune ones about half (two of
    # four levels at 5 %), nothing dropped
    by = {w["name"]: w for w in d["weak_regimes"]}
    assert by["weak_dense_all_channels"]["exchanged_bytes"] == by["weak_dense_all_channels"]["dense_bytes"] == 4 * 4096 * 2 * 4
    for name in ("weak_post_prune_rgb", "weak_post_prune_all_channels"):
        assert by[name]["sparse_sync"] == "bounded" and by[name]["whole_levels"] == 2 and by[name]["dropped_rows"] == 0
        assert by[name]["exchanged_bytes"] < 0.4 * by[name]["dense_bytes"], by[name]         # bf16 messages: 2 whole levels + the slots of two 5 % levels
    assert small["weak_regimes"]["weak_post_prune_rgb"]["exchanged_bytes"] == by["weak_post_prune_rgb"]["exchanged_bytes"]
    # the driver's largest launch: 8 ranks (gloo here), with the fp32 all-reduce and with bf16 messages + fp32 accumulation
    for extra in (["--grad-sync", "fp32"], ["--grad-sync", "bf16"]):
        r8 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--dry-run", "--steps", "2", "--warmup", "1"] + extra,
                            capture_output=True, text=True, env=dict(env, OMP_NUM_THREADS="1"), timeout=900)
        assert r8.returncode == 0, r8.stderr[-2000:]
        d8 = json.loads([l for l in r8.stdout.splitlines() if l.startswith("{")][0])
        assert d8["n_gpus"] == 8 and d8["rccl_ranks_seen"] == 8 and d8["grad_sync"] == extra[1]
    # N = 1 dry run: no spawn, same schema
    r1 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--dry-run", "--steps", "2", "--warmup", "0"],
                        capture_output=True, text=True, env=env, timeout=600)
    assert r1.returncode == 0 and json.loads(r1.stdout.strip().splitlines()[-1])["n_gpus"] == 1
    # a rank count that does not match --gpus refuses to report
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True,
                        env=env2, timeout=600)
    assert r2.returncode != 0 and not [l for l in r2.stdout.splitlines() if l.startswith("{")]


def test_bench_result_line_stays_small_and_parseable():
    """round 5's result line grew to 32 KB and the driver recorded `parsed: null`.  bench.compact_line() on the largest record this
    repository has produced (the committed round-5 run: 32 KB, every optional block present) must stay under 4 KB, survive a JSON round trip and keep the
    contract's keys with `roofline` and `cpu_baseline` as flat objects; an absurdly large record must still come out under the budget."""
    import json
    import os
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    full = json.load(open(os.path.join(root, "profiles", "r05h_bench_default_run.json")))
    assert len(json.dumps(full)) > 30000
    line = bench.compact_line(full, "bench_detail.json")
    text = json.dumps(line)
    assert len(text) < 4096 and json.loads(text) == line and "\n" not in text
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["value"] == full["value"] and line["ms_per_step"] == full["ms_per_step"] and line["config"]["workload"].startswith("BUP20-shaped")
    rf, cb = line["roofline"], line["cpu_baseline"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(rf) and all(not isinstance(v, (dict, list)) for v in rf.values())
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert set(cb) == {"value", "unit", "cores", "cores_available", "kind", "sample"} and cb["kind"] == "port" and len(cb["sample"]) <= 300
    assert line["mfma_util"] == {"frac": full["mfma_util"]["frac"]} and line["schedule_weighted"]["ms_per_step"] == full["schedule_weighted"]["ms_per_step"]
    assert len(line["configs"]) == len(full["configs"]) and all(isinstance(v, (int, float)) for v in line["configs"].values())
    fat = dict(full, configs=[dict(name="configs[9]: %d %s" % (i, "x" * 200), ms_per_step=1.0) for i in range(400)], kernels={("k%d" % i): dict(ms_per_step=1.0) for i in range(400)})
    slim = bench.compact_line(fat, "bench_detail.json")
    assert len(json.dumps(slim)) <= bench.LINE_BUDGET and "roofline" in slim and "cpu_baseline" in slim and slim["value"] == full["value"]


def test_nef_option_decoder_widths_and_guards():
    """Decoder input widths per panoptic_features_type / multiscale_type (pc_nerf/panoptic_nef.py:78-105) and the option guards
    (panoptic_delta_nef.py:177, :234) - host logic only."""
    import pytest
    import pagnerf_amd
    kw = dict(grid_type="PermutoGrid", feature_dim=2, num_lods=24, num_classes=6, num_instances=200, capacity_log_2=8, delta_capacity_log_2=8)
    for t, want, has_delta in ((None, 48, True), ("delta", 48, True), ("separate", 48, True), ("appearance", 48, False),
                               ("pos_encoding", 27, False), ("position", 3, False)):
        nef = pagnerf_amd.PanopticDeltaNeF(panoptic_features_type=t, **kw)
        assert nef.decoder_semantics.input_dim == want and nef.decoder_inst.input_dim == want and nef.decoder_density.input_dim == 48
        assert hasattr(nef, "delta_grid") == has_delta
    nef = pagnerf_amd.PanopticDeltaNeF(multiscale_type="sum", **kw)
    assert nef.decoder_density.input_dim == 2 and nef.decoder_inst.input_dim == 2 and nef._grouped() is None
    with pytest.raises(ValueError):
        pagnerf_amd.PanopticDeltaNeF(panoptic_features_type="nope", **kw)
    with pytest.raises(NotImplementedError):
        pagnerf_amd.PanopticDeltaNeF(position_input=True, **kw)
    with pytest.raises(NotImplementedError):
        pagnerf_amd.PanopticDeltaNeF(multiscale_type="max", **kw)


def test_bench_byte_model_matches_design_table():
    """bench.algorithmic_model() - the per-launch algorithmic bytes the `kernels` block of the bench line is computed from - against the
    table of DESIGN.md section 5 (production path: bf16 features in the XCD8 layout, fused backward kernels), and against the committed
    PMC traffic of profiles/ within 1.3x for every streaming decoder kernel."""
    import json
    import os
    import bench
    M, N = 4096 * 512, 4096
    m = bench.algorithmic_model("permuto", M, N, {"rgb", "depth", "semantics", "inst_embedding"}, 24, 2, 4, True)
    per = {k: v["bytes"] / M for k, v in m.items()}
    # 512 samples per ray: the 200-way head's forward is the one-launch form (decoder + per-ray sum, pag_mlp_fwd_args.composite) - no pag_head_composite_fwd
    assert m["pag_mlp_fwd"]["parts"] == {"density": 160, "colour": 52, "inst_once+sem": 280} and m["pag_mlp_fwd"]["bytes"] == 492 * M + N * 800
    assert "pag_head_composite_fwd" not in m
    two = bench.algorithmic_model("permuto", M, N, {"rgb", "depth", "semantics", "inst_embedding"}, 24, 2, 4, True, head_once=False)
    assert two["pag_mlp_fwd"]["parts"] == {"density": 160, "colour": 52, "inst_stats+sem": 276} and two["pag_mlp_fwd"]["bytes"] == 488 * M
    assert two["pag_head_composite_fwd"]["bytes"] == M * 140 + N * 800
    assert m["pag_mlp_bwd"]["parts"] == {"density": 288, "colour": 88, "inst_stage_A": 264, "inst_stage_B+sem": 396} and per["pag_mlp_bwd"] == 1036
    assert per["pag_permuto_encode_fwd"] == 876 and per["pag_permuto_encode_fwd_add"] == 972 and per["pag_permuto_encode_bwd_set"] == 2 * 1644
    assert m["pag_mlp_fwd"]["flops"] == 2 * M * 34560 and m["pag_mlp_bwd"]["flops"] == 2 * M * (34560 + 32832)
    assert m["pag_adam_step"]["bytes"] == 28 * (2 * 24 * 262144 * 2 + 35169)          # 16 B read + 12 B written per fp32 parameter: 705.6 MB per step
    rgb = bench.algorithmic_model("permuto", M, N, {"rgb"}, 24, 2, 4, True)
    # rgb / rgb + depth steps: the delta table and the panoptic heads carry no gradient and are not stepped (hbm_frac 1.39 came from counting both)
    assert rgb["pag_adam_step"]["bytes"] == 28 * (24 * 262144 * 2 + 35169 - bench.DECODER_PARAMS_PANOPTIC) and bench.DECODER_PARAMS_PANOPTIC == 23822
    assert rgb["pag_permuto_encode_bwd_rays"]["bytes"] == M * (12 + 96 + 768 + 8) + N * 24
    assert rgb["pag_mlp_fwd"]["bytes"] / M == 212 and rgb["pag_permuto_encode_bwd_set"]["bytes"] / M == 1644 and "pag_head_composite_fwd" not in rgb
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pdir = os.path.join(root, "profiles")
    newest = sorted(f for f in os.listdir(pdir) if f.endswith("_pmc_traffic_per_launch.json"))[-1]
    blob = json.load(open(os.path.join(pdir, newest)))
    for entry in ("pag_mlp_fwd", "pag_mlp_bwd", "pag_composite_fwd"):
        pmc = bench.pmc_bytes_per_step(blob, entry, 1)
        assert pmc is not None and 1 / 1.3 < pmc / m[entry]["bytes"] < 1.3, (entry, pmc, m[entry]["bytes"])


def test_xcd8_layout_is_the_snake_permutation():
    """PAG_LAYOUT_XCD8 (C ABI 6): element j*F + f of group g holds level 8j + g for even j, 8j + 7 - g for odd j.  The host-side map
    (what tests and callers unpack the tensor with) must be a bijection onto the L*F feature columns, pad everything else, and put
    the 24-level grid's levels where DESIGN.md section 4.1 says (g, 15 - g, 16 + g)."""
    from pagnerf_amd import ops
    assert [ops.xcd8_level(g, 0) for g in range(8)] == list(range(8))
    assert [ops.xcd8_level(g, 1) for g in range(8)] == list(range(15, 7, -1))
    assert [ops.xcd8_level(g, 2) for g in range(8)] == list(range(16, 24))
    for L_, F_ in ((24, 2), (16, 2), (20, 2), (16, 4), (12, 4), (7, 2), (1, 1), (32, 2), (64, 1), (8, 8)):
        assert ops.xcd8_supported(L_, F_)
        cols = ops.xcd8_columns(L_, F_)
        assert len(cols) == 64
        real = sorted(c for c in cols if c >= 0)
        assert real == list(range(L_ * F_)), (L_, F_)
        for pos, c in enumerate(cols):
            if c >= 0:
                g, e = pos >> 3, pos & 7
                assert c == ops.xcd8_level(g, e // F_) * F_ + e % F_
    assert not ops.xcd8_supported(33, 2) and not ops.xcd8_supported(24, 4)
    # the cheap / expensive pairing the order exists for: every group of the 24-level grid sums to the same level total
    sums = {sum(ops.xcd8_level(g, j) for j in range(2)) for g in range(8)}
    assert sums == {15}


def test_graph_capacity_buckets_are_geometric_with_hysteresis():
    """pagnerf_amd/graphs.py::_State.capacity: every new capacity is a capture (two warm-up steps, a private memory pool), so batch-to-batch
    noise must not move it: geometric buckets (<= 1/32 of a power of two of filler), immediate growth, shrink only after SHRINK_WINDOW
    steps that would all fit the smaller bucket; at most MAX_BUCKETS captures stay alive per configuration."""
    from pagnerf_amd import graphs

    class Buf:
        k, cap = 1, 4096 * 512
    st = graphs._State()
    rng = np.random.default_rng(0)
    caps = set()
    base = 1_600_000
    for it in range(400):                       # +-0.3 % noise and a slow 3 % drift (what follows a prune): a handful of capacities (the 8192-sample
                                                # granule alone gave ~ 20 here), not one every few steps
        st.see(int(base * (1 + 0.03 * it / 400) * (1 + 0.003 * rng.standard_normal())))
        caps.add(st.capacity(Buf))
    assert len(caps) <= 3, sorted(caps)
    cap = st.cap
    assert cap % graphs.GRANULE == 0 and cap <= Buf.cap
    assert cap >= max(st.counts) * graphs.HEADROOM and cap <= max(st.long) * graphs.HEADROOM * (1 + 1 / graphs.CAP_STEPS) + graphs.GRANULE
    # growth is immediate
    st.see(1_900_000)
    assert st.capacity(Buf) >= 1_900_000 * graphs.HEADROOM
    grown = st.cap
    # a smaller batch does not shrink it until a whole window has passed
    for it in range(graphs.SHRINK_WINDOW - 1):
        st.see(400_000)
        assert st.capacity(Buf) == grown
    st.see(400_000)
    small = st.capacity(Buf)
    assert small < grown and small >= 400_000 * graphs.HEADROOM and small <= 400_000 * 1.02 * (1 + 1 / graphs.CAP_STEPS) + graphs.GRANULE
    # never above the march buffers' upper bound (dense occupancy: M = N * S exactly)
    st.see(Buf.cap)
    assert st.capacity(Buf) == Buf.cap
    # voxel mode: multiples of GRANULE * k
    class Buf2:
        k, cap = 2, 10 ** 9
    st2 = graphs._State()
    st2.see(325_001)
    assert st2.capacity(Buf2) % (graphs.GRANULE * 2) == 0
    # LRU of captured capacities
    made = []
    for c in (1, 2, 3, 2, 4):
        st.bucket(c, lambda c=c: made.append(c) or ("graph", c))
    assert list(st.buckets) == [2, 4] and made == [1, 2, 3, 4]


def test_adam_is_a_torch_adam_with_the_same_interface():
    """pagnerf_amd.optim.Adam: constructor / param_groups / state_dict of torch.optim.Adam (config_parser.py:667-673 builds `optim_cls(params,
    **optim_params)`); CPU parameters take torch's own step (the kernel needs GPU tensors)."""
    import pagnerf_amd
    p = torch.nn.Parameter(torch.ones(5))
    q = torch.nn.Parameter(torch.ones(5))
    oa = pagnerf_amd.optim.Adam([dict(params=[p], lr=0.1)], lr=1e-3, eps=1e-15, fused=True)
    ob = torch.optim.Adam([dict(params=[q], lr=0.1)], lr=1e-3, eps=1e-15)
    assert isinstance(oa, torch.optim.Adam) and oa.defaults["eps"] == 1e-15 and oa.param_groups[0]["lr"] == 0.1
    for _ in range(3):
        p.grad, q.grad = torch.full((5,), 0.5), torch.full((5,), 0.5)
        oa.step()
        ob.step()
    assert torch.equal(p.detach(), q.detach())
    sd = oa.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and float(sd["state"][0]["step"]) == 3.0
    ob.load_state_dict(sd)
    sched = torch.optim.lr_scheduler.StepLR(oa, step_size=1, gamma=0.1)
    sched.step()
    assert abs(oa.param_groups[0]["lr"] - 0.01) < 1e-12


def test_adam_group_gate_and_hooks_fire_once():
    """Groups the kernel's arithmetic does not cover are refused by the gate (amsgrad, maximize, decoupled weight decay with a non-zero decay,
    tensor-valued lr / betas) and take torch's step - entered below its hook wrapper, so step hooks fire once per step()."""
    import pagnerf_amd
    ok = pagnerf_amd.optim.Adam._group_ok
    base = dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-15, weight_decay=0.0, amsgrad=False, maximize=False, capturable=False, differentiable=False,
                decoupled_weight_decay=False)
    assert ok(base) and ok(dict(base, weight_decay=0.1)) and ok(dict(base, decoupled_weight_decay=True))        # decoupled with wd = 0 is plain Adam
    for bad in (dict(amsgrad=True), dict(maximize=True), dict(decoupled_weight_decay=True, weight_decay=0.1), dict(lr=torch.tensor(1e-3)),
                dict(betas=(torch.tensor(0.9), 0.999)), dict(capturable=True)):
        assert not ok(dict(base, **bad)), bad
    torch.optim.Adam([torch.nn.Parameter(torch.ones(1))])          # a plain torch Adam exists: torch.optim.Adam.step is hook-wrapped from here on
    p, q = torch.nn.Parameter(torch.ones(5)), torch.nn.Parameter(torch.ones(5))
    oa = pagnerf_amd.optim.Adam([p], lr=0.1, weight_decay=0.1, decoupled_weight_decay=True)
    ob = torch.optim.Adam([q], lr=0.1, weight_decay=0.1, decoupled_weight_decay=True)
    fired = []
    oa.register_step_pre_hook(lambda *a: fired.append("pre"))
    oa.register_step_post_hook(lambda *a: fired.append("post"))
    for _ in range(3):
        p.grad, q.grad = torch.full((5,), 0.5), torch.full((5,), 0.5)
        oa.step()
        ob.step()
    assert fired == ["pre", "post"] * 3 and torch.equal(p.detach(), q.detach())
    assert oa.param_groups[0]["params"][0] is p          # the group list the caller sees is the original one again


def test_graph_key_parameter_signature_without_a_module_walk():
    """GraphRunner._param_sig: the (data_ptr, requires_grad) signature of the nef's parameters read from remembered (module, name) slots - a
    frozen parameter, a replaced Parameter object and a different nef all change it; the same nef gives the same signature."""
    import gc
    import pagnerf_amd
    from pagnerf_amd.graphs import GraphRunner

    def make():
        nef = pagnerf_amd.PanopticDeltaNeF(grid_type="PermutoGrid", num_lods=4, feature_dim=2, num_classes=3, num_instances=5, capacity_log_2=6,
                                           delta_capacity_log_2=6, blas_level=2)
        nef.grid.init_from_scales()
        nef.delta_grid.init_from_scales()
        return nef
    r = GraphRunner()
    nef = make()
    a = r._param_sig(nef)
    assert a == r._param_sig(nef) and len(a) == len(list(nef.parameters()))
    nef.decoder_density.lout.weight.requires_grad_(False)
    b = r._param_sig(nef)
    assert b != a
    nef.grid.init_from_scales()                      # replaces nef.grid.tables by a new Parameter: seen without a rescan
    c = r._param_sig(nef)
    assert c != b
    other = make()
    assert r._param_sig(other) != c and r._param_sig(nef) == c
    del other
    gc.collect()
    assert r._param_sig(nef) == c
    # a module attached between two periodic rescans is seen at the next step (structural fingerprint), as is one swapped in for another
    nef.extra_head = torch.nn.Linear(3, 3)
    d = r._param_sig(nef)
    assert len(d) == len(c) + 2
    nef.extra_head = torch.nn.Linear(3, 3)
    e = r._param_sig(nef)
    assert len(e) == len(d) and e != d
    # two nefs alternating on one runner keep a slot cache each
    other = make()
    for _ in range(3):
        r._param_sig(other)
        r._param_sig(nef)
    assert len(r._slot_caches) == 2 and r._slot_caches[id(nef)][2] < r.PARAM_RESCAN - 2


def test_batch_render_keeps_channels_that_appear_in_later_chunks():
    """pagnerf_amd.batch_render joins the per-chunk buffers channel by channel over ALL chunks (trainer.py:648 `rb += ...` keeps a channel
    that is None in the first chunk and a tensor in a later one)."""
    import pagnerf_amd

    class P:
        def __init__(self):
            self.n = 0

        def __call__(self, rays, lod_idx=None, channels=None):
            self.n += 1
            k = rays.origins.shape[0]
            return pagnerf_amd.RenderBuffer(rgb=torch.full((k, 3), float(self.n)), depth=None if self.n == 1 else torch.full((k, 1), float(self.n)),
                                            reg=torch.tensor(float(self.n)))
    rays = pagnerf_amd.Rays(torch.zeros(10, 3), torch.zeros(10, 3))
    rb = pagnerf_amd.batch_render(P(), rays, render_batch=4)
    assert rb.rgb.shape == (10, 3) and rb.depth.shape == (6, 1) and float(rb.reg) == 1.0
    assert torch.equal(rb.rgb[:, 0], torch.tensor([1.0] * 4 + [2.0] * 4 + [3.0] * 2))


def test_segment_consistency_regularizer_host_logic_vs_reference_golden():
    """pagnerf_amd.loss.segment_consistency_regularizer is plain tensor ops (no C-ABI call): on CPU tensors it reproduces the value and gradient the
    reference's loss/regularizers.py:5-35 gave for the g6 batch (the GPU run of the same check: tests/test_gpu_loss.py)."""
    import numpy as np
    from conftest import golden
    from pagnerf_amd import loss as pl
    g = golden("g6_reg.npz")
    x = (torch.from_numpy(g["seg_prob"]) + 1e-27).requires_grad_(True)
    val = pl.segment_consistency_regularizer(x, torch.from_numpy(g["seg_labels"]))
    val.backward()
    np.testing.assert_allclose(float(val.detach()), float(g["seg_reg"]), rtol=1e-5)
    np.testing.assert_allclose(x.grad.numpy(), g["seg_reg_grad"], rtol=1e-5, atol=0)


def test_tracer_defaults_to_graphs(monkeypatch):
    """Drop-in default: a tracer constructed the way the reference constructs it - no `use_graphs` keyword - takes the HIP-graph
    path for training traces; PAG_GRAPHS=0 / use_graphs=False switch it off, "static" selects the uncaptured static-buffer form."""
    import pagnerf_amd
    monkeypatch.delenv("PAG_GRAPHS", raising=False)
    assert pagnerf_amd.PanopticPackedRFTracer(raymarch_type="ray", num_steps=512).use_graphs is True
    assert pagnerf_amd.PanopticPackedRFTracer(use_graphs=False).use_graphs is False
    assert pagnerf_amd.PanopticPackedRFTracer(use_graphs="static").use_graphs == "static"
    monkeypatch.setenv("PAG_GRAPHS", "0")
    assert pagnerf_amd.PanopticPackedRFTracer().use_graphs is False
    monkeypatch.setenv("PAG_GRAPHS", "static")
    assert pagnerf_amd.PanopticPackedRFTracer().use_graphs == "static"
