"""ctypes binding of libpagnerf_hip.so (include/pagnerf_hip.h).

The product path has NO CPU fallback: if the library is missing or a GPU tensor is not handed
in, calls raise.  Import of this module alone never touches the GPU.
"""
import ctypes
import os

import torch

from . import build as _build

F32, F16, BF16 = 0, 1, 2
ACT_NONE, ACT_SIGMOID, ACT_SOFTMAX = 0, 1, 2
MLP_MFMA_BF16, MLP_FP32 = 0, 1
BG_BLACK, BG_WHITE = 0, 1
LAYOUT_STRIDED, LAYOUT_XCD8 = 0, 1
ENC_HALF_COORDS = 1
ABI_VERSION = 14
MLP_FUSED_WIDE_MAX_M = 1 << 24      # PAG_MLP_FUSED_WIDE_MAX_M

_DT = {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16}

c_i64, c_i32, c_u32, c_f32, c_vp = ctypes.c_int64, ctypes.c_int, ctypes.c_uint32, ctypes.c_float, ctypes.c_void_p
c_fp = ctypes.POINTER(ctypes.c_float)


class HeadCompositeArgs(ctypes.Structure):
    """pag_head_composite_args (include/pagnerf_hip.h)."""
    _fields_ = [("pack_start", c_vp), ("ray_of_pack", c_vp), ("P", c_i64), ("weights", c_vp), ("alpha", c_vp), ("out", c_vp), ("n_samples", c_i64)]


class MlpFwdArgs(ctypes.Structure):
    pass


MlpFwdArgs._fields_ = [("x1", c_vp), ("x1_dtype", c_i32), ("k1", c_i32),
                ("x1_layout", c_i32), ("x1_levels", c_i32), ("x1_feats", c_i32),
                ("x2", c_vp), ("k2p", c_i32), ("x2_index", c_vp),
                ("in_dim", c_i32), ("n_layers", c_i32), ("out_dim", c_i32),
                ("W", c_vp * 3), ("b", c_vp * 3),
                ("out_act", c_i32),
                ("out", c_vp), ("out_dtype", c_i32),
                ("hidden_save", c_vp * 2),
                ("mode", c_i32),
                ("softmax_stats", c_vp), ("x1_col0_relu", c_vp), ("pair", ctypes.POINTER(MlpFwdArgs)),
                ("composite", ctypes.POINTER(HeadCompositeArgs)), ("x1_producer", ctypes.POINTER(MlpFwdArgs))]


class WgradLayer(ctypes.Structure):
    """pag_wgrad_layer (include/pagnerf_hip.h)."""
    _fields_ = [("dz", c_vp), ("dz_cols", c_i32), ("n_out", c_i32),
                ("a1", c_vp), ("a1_dtype", c_i32), ("a1_layout", c_i32), ("k1", c_i32),
                ("a2", c_vp), ("k2p", c_i32), ("a2_index", c_vp), ("n_in", c_i32),
                ("slabs", c_vp), ("n_blocks", c_i32), ("a1_levels", c_i32), ("a1_feats", c_i32),
                ("dW", c_vp), ("db", c_vp)]


class MlpBwdArgs(ctypes.Structure):
    pass


MlpBwdArgs._fields_ = [("grad_out", c_vp), ("out", c_vp), ("out_dtype", c_i32), ("out_act", c_i32),
                ("k1", c_i32), ("in_dim", c_i32), ("n_layers", c_i32), ("out_dim", c_i32),
                ("x1_layout", c_i32), ("x1_levels", c_i32), ("x1_feats", c_i32),
                ("W", c_vp * 3),
                ("hidden_save", c_vp * 2),
                ("dz", c_vp * 3),
                ("dx1", c_vp), ("dx1_dtype", c_i32),
                ("mode", c_i32),
                ("g_ray", c_vp), ("g_scale", c_vp), ("g_index", c_vp),
                ("softmax_stats", c_vp), ("b_last", c_vp), ("dx1_accumulate", c_i32), ("dx1_col0_add", c_vp),
                ("dx1_col0_gate", c_vp), ("g_ray_scale", c_vp),
                ("x1", c_vp), ("x1_dtype", c_i32), ("x2", c_vp), ("k2p", c_i32), ("x2_index", c_vp),
                ("wgrad_workspace", c_vp), ("wgrad_workspace_bytes", c_i64),
                ("dW", c_vp * 3), ("db", c_vp * 3), ("b", c_vp * 3),
                ("pair", ctypes.POINTER(MlpBwdArgs)), ("dz0_slots", c_vp)]


_SIGS = {
    "pag_abi_version": (c_i32, []),
    "pag_last_error_string": (ctypes.c_char_p, []),
    "pag_hash_encode_fwd": (c_i32, [c_vp, c_i64, c_vp, c_i32, c_i32, c_i32, c_i32, c_fp, c_fp, c_vp, c_i32, c_i64, c_i64, c_i32, c_i32, c_vp]),
    "pag_hash_encode_bwd": (c_i32, [c_vp, c_i64, c_vp, c_i32, c_i64, c_i64, c_i32, c_i32, c_i32, c_i32, c_fp, c_fp, c_vp, c_vp, c_i64, c_i32, c_vp]),
    "pag_hash_encode_bwd_set": (c_i32, [c_vp, c_i64, c_vp, c_i32, c_i64, c_i64, c_i32, c_i32, c_i32, c_i32, c_fp, c_fp, c_vp, c_vp, c_i64, c_i32, c_vp]),
    "pag_permuto_encode_fwd": (c_i32, [c_vp, c_i64, c_vp, c_i32, c_i32, c_i32, c_u32, c_fp, c_fp, c_fp, c_vp, c_i32, c_i64, c_i64, c_i32, c_i32, c_vp]),
    "pag_permuto_encode_bwd": (c_i32, [c_vp, c_i64, c_vp, c_i32, c_i64, c_i64, c_i32, c_i32, c_i32, c_u32, c_fp, c_fp, c_fp, c_vp, c_vp, c_i64, c_i32, c_vp]),
    "pag_permuto_encode_bwd_set": (c_i32, [c_vp, c_i64, c_vp, c_i32, c_i64, c_i64, c_i32, c_i32, c_i32, c_u32, c_fp, c_fp, c_fp, c_vp, c_vp, c_i64, c_i32, c_vp]),
    "pag_hash_encode_bwd_xyz": (c_i32, [c_vp, c_i64, c_vp, c_i32, c_vp, c_i32, c_i64, c_i64, c_i32, c_i32, c_i32, c_i32, c_fp, c_fp, c_vp, c_vp, c_i64, c_i32, c_vp]),
    "pag_permuto_encode_bwd_xyz": (c_i32, [c_vp, c_i64, c_vp, c_i32, c_vp, c_i32, c_i64, c_i64, c_i32, c_i32, c_i32, c_u32, c_fp, c_fp, c_fp, c_vp, c_vp, c_i64, c_i32, c_vp]),
    "pag_hash_encode_fwd_add": (c_i32, [c_vp, c_i64, c_vp, c_i32, c_i32, c_i32, c_i32, c_fp, c_fp, c_vp, c_vp, c_i32, c_vp]),
    "pag_permuto_encode_fwd_add": (c_i32, [c_vp, c_i64, c_vp, c_i32, c_i32, c_i32, c_u32, c_fp, c_fp, c_fp, c_vp, c_vp, c_i32, c_vp]),
    "pag_encode_bwd_workspace_bytes": (c_i64, [c_i64, c_i32, c_i32, c_i32, c_i64]),
    "pag_mlp_fwd": (c_i32, [ctypes.POINTER(MlpFwdArgs), c_i64, c_vp]),
    "pag_mlp_fwd_pair_supported": (c_i32, [ctypes.POINTER(MlpFwdArgs), ctypes.POINTER(MlpFwdArgs)]),
    "pag_mlp_fwd_composite_supported": (c_i32, [ctypes.POINTER(MlpFwdArgs), c_i64]),
    "pag_mlp_fwd_producer_supported": (c_i32, [ctypes.POINTER(MlpFwdArgs), ctypes.POINTER(MlpFwdArgs), c_i64]),
    "pag_mlp_bwd": (c_i32, [ctypes.POINTER(MlpBwdArgs), c_i64, c_vp]),
    "pag_mlp_bwd_fused_supported": (c_i32, [ctypes.POINTER(MlpBwdArgs)]),
    "pag_mlp_bwd_fused_workspace_bytes": (c_i64, [ctypes.POINTER(MlpBwdArgs), c_i64]),
    "pag_mlp_bwd_pair_supported": (c_i32, [ctypes.POINTER(MlpBwdArgs), ctypes.POINTER(MlpBwdArgs)]),
    "pag_head_composite_fwd": (c_i32, [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "pag_mlp_wgrad_blocks": (c_i32, [c_i64]),
    "pag_mlp_wgrad_batch": (c_i32, [ctypes.POINTER(WgradLayer), c_i32, c_i64, c_vp]),
    "pag_mlp_wgrad_finish": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "pag_mlp_wgrad": (c_i32, [c_vp, c_i32, c_i32, c_vp, c_i32, c_i32, c_i32, c_vp, c_i32, c_vp, c_i32, c_vp, c_i32, c_i64, c_vp]),
    "pag_raymarch_count": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_f32, c_f32, c_vp, c_i32, c_vp, c_vp]),
    "pag_pack_offsets": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp]),
    "pag_pad_packed": (c_i32, [c_vp, c_i64, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "pag_view_embed": (c_i32, [c_vp, c_i64, c_i32, c_i32, c_vp, c_vp]),
    "pag_raymarch_pack": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_f32, c_f32, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "pag_raymarch_voxel_count": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_f32, c_f32, c_vp, c_vp, c_i32, c_f32, c_vp, c_vp]),
    "pag_raymarch_voxel_pack": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_f32, c_f32, c_vp, c_vp, c_i32, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                        c_vp, c_vp, c_vp]),
    "pag_raymarch_voxel_nugget_capacity": (c_i64, [c_i32]),
    "pag_raymarch_voxel_count_nuggets": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_f32, c_f32, c_vp, c_vp, c_i32, c_f32, c_vp, c_vp, c_vp, c_vp]),
    "pag_raymarch_voxel_pack_nuggets": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "pag_ray_sample_grad": (c_i32, [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp]),
    "pag_affine_xcd8_fwd": (c_i32, [c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp]),
    "pag_affine_xcd8_bwd_dx": (c_i32, [c_vp, c_i64, c_i32, c_i32, c_vp, c_i32, c_i32, c_vp, c_vp]),
    "pag_occupancy_coarse_bytes": (c_i64, [c_i32]),
    "pag_occupancy_coarse": (c_i32, [c_vp, c_i32, c_vp, c_vp]),
    "pag_occupancy_update": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i64, c_f32, c_f32, c_vp]),
    "pag_label_sums": (c_i32, [c_vp, c_i32, c_i64, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp]),
    "pag_assign_cost": (c_i32, [c_vp, c_i32, c_i64, c_i64, c_i64, c_i32, c_i32, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.c_float, ctypes.c_float, c_i32,
                                c_vp, c_vp, c_vp, c_vp]),
    "pag_sparse_rows_mask": (c_i32, [c_vp, c_i32, c_i64, c_i32, c_vp, c_vp]),
    "pag_sparse_rows_plan": (c_i32, [c_vp, c_i32, c_i64, c_vp, c_vp, c_vp, c_vp]),
    "pag_sparse_rows_pack": (c_i32, [c_vp, c_i32, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "pag_sparse_rows_unpack": (c_i32, [c_vp, c_i32, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "pag_assign_solve": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "pag_assign_nll_fwd": (c_i32, [c_vp, c_i32, c_i64, c_i64, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "pag_assign_nll_bwd": (c_i32, [c_vp, c_i32, c_i64, c_i64, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "pag_render_loss_workspace_bytes": (c_i64, []),
    "pag_render_loss_fwd": (c_i32, [c_vp, c_vp, c_i64, c_f32, c_vp, c_i32, c_vp, c_vp, c_f32, c_f32, c_i32, c_vp, c_i32, c_vp, c_vp, c_f32, c_f32, c_i32, c_f32, c_vp, c_vp, c_vp]),
    "pag_render_loss_bwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_f32, c_vp, c_i32, c_vp, c_vp, c_f32, c_f32, c_i32, c_vp, c_i32, c_vp, c_vp, c_f32, c_f32, c_i32, c_f32, c_vp, c_vp, c_vp, c_vp]),
    "pag_composite_fwd": (c_i32, [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "pag_composite_bwd": (c_i32, [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "pag_composite_feats_fwd": (c_i32, [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp]),
    "pag_composite_feats_bwd": (c_i32, [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i32, c_vp, c_i32, c_vp]),
    "pag_pack_offsets_pad": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "pag_copy_batch": (c_i32, [c_i32, c_vp, c_vp, c_vp, c_vp]),
    "pag_adam_step": (c_i32, [c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                              ctypes.c_double, c_i64, c_vp]),
    "pag_pose_rays_fwd": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "pag_pose_rays_bwd_workspace_bytes": (c_i64, [c_i64]),
    "pag_pose_rays_bwd": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "pag_view_embed_bwd": (c_i32, [c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "pag_pose_points": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp]),
    "pag_segment_reg_workspace_bytes": (c_i64, [c_i32, c_i64]),
    "pag_segment_reg_fwd": (c_i32, [c_vp, c_i32, c_i64, c_i64, c_i64, c_i32, ctypes.c_float, c_vp, c_vp, c_i64, c_vp, c_vp]),
    "pag_segment_reg_bwd": (c_i32, [c_vp, c_i32, c_i64, c_i64, c_i64, c_i32, ctypes.c_float, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "pag_mlp_dz0_slots_bytes": (c_i64, [c_i64, c_i64]),
    "pag_mlp_dz0_slots_sum": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp]),
    "pag_encode_bwd_rays_workspace_bytes": (c_i64, [c_i64, c_i64]),
    "pag_hash_encode_bwd_rays": (c_i32, [c_vp, c_i64, c_vp, c_i32, c_vp, c_i32, c_i64, c_i64, c_i32, c_i32, c_i32, c_i32, c_fp, c_fp, c_vp, c_vp, c_vp, c_i64,
                                         c_vp, c_vp, c_i64, c_i32, c_vp]),
    "pag_permuto_encode_bwd_rays": (c_i32, [c_vp, c_i64, c_vp, c_i32, c_vp, c_i32, c_i64, c_i64, c_i32, c_i32, c_i32, c_u32, c_fp, c_fp, c_fp, c_vp, c_vp, c_vp,
                                            c_i64, c_vp, c_vp, c_i64, c_i32, c_vp]),
}

EXPORTS = tuple(_SIGS)
_lib = None


def library_path():
    """The in-tree library; PAG_LIB_VARIANT=<tag> (kernel experiments only, scripts/build_variant.sh) loads lib/libpagnerf_hip_<tag>.so -
    the same sources built with extra -D flags - so that variants can be A/B-timed on one box without rebuilding there."""
    tag = os.environ.get("PAG_LIB_VARIANT")
    if tag:
        return os.path.join(os.path.dirname(_build.LIB), "libpagnerf_hip_%s.so" % tag)
    return _build.LIB


def load():
    """dlopen the C-ABI library (building is __graft_entry__.build()'s job); raises if absent."""
    global _lib
    if _lib is None:
        path = library_path()
        if not os.path.exists(path):
            raise RuntimeError("libpagnerf_hip.so not found at %s - run `python -m pagnerf_amd.build` "
                               "(there is no CPU fallback for the HIP path)" % path)
        lib = ctypes.CDLL(path)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        if lib.pag_abi_version() != ABI_VERSION:
            raise RuntimeError("libpagnerf_hip.so ABI version mismatch")
        _lib = lib
    return _lib


def check(rc, name):
    if rc != 0:
        msg = load().pag_last_error_string().decode("utf-8", "replace")
        raise RuntimeError("%s failed (%d): %s" % (name, rc, msg))


def dtype_code(t):
    return _DT[t.dtype]


def ptr(t):
    """Device pointer of a contiguous GPU tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("pagnerf_amd: expected a GPU tensor (the HIP path has no CPU fallback), got %s" % t.device)
    if not t.is_contiguous():
        raise RuntimeError("pagnerf_amd: tensor must be contiguous")
    return t.data_ptr()


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """The raw handle of torch's current stream on the current device.  torch.cuda.current_stream() builds a Stream object per call (~10 us: a tenth of
    the host time of a forward-only trace, which asks nine times); the C-level getter returns the same handle in well under a microsecond."""
    if _RAW_STREAM is not None:
        return _RAW_STREAM(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def host_floats(values):
    """python/numpy/torch-CPU float sequence -> ctypes float array (kept alive by the caller)."""
    if values is None:
        return None
    if isinstance(values, torch.Tensor):
        values = values.detach().cpu().float().reshape(-1).tolist()
    else:
        import numpy as np
        values = np.asarray(values, dtype=np.float32).reshape(-1).tolist()
    return (ctypes.c_float * len(values))(*values)
