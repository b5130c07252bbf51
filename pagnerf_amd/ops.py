"""torch.autograd wrappers around the C-ABI kernels (include/pagnerf_hip.h).

Everything here runs on GPU tensors only; there is no CPU fallback (the oracle under oracle/
is test infrastructure and is never imported from this package).
"""
import ctypes
import os
import threading
import time

import torch

from . import _lib as L


# Optional per-entry-point device timing (bench.py): HIP events recorded on the launch stream
# (torch's current stream, which is the stream every kernel is launched on) around each C-ABI call.
_PROFILE = None


_PROFILE_ONLY = None


def profile_start(only=None):
    """only: optional set of entry-point names - events are then recorded around those calls alone (two event records per
    call cost host time and a marker packet on the stream; ~50 of them per training step are measurable)."""
    global _PROFILE, _PROFILE_ONLY
    _PROFILE = {}
    _PROFILE_ONLY = set(only) if only is not None else None


def profile_stop():
    """-> {entry point: [milliseconds per call]} (synchronises)."""
    global _PROFILE
    prof, _PROFILE = _PROFILE, None
    torch.cuda.synchronize()
    return {k: [a.elapsed_time(b) for a, b in v] for k, v in (prof or {}).items()}


def _call(name, *args):
    fn = getattr(L.load(), name)
    if _PROFILE is None or (_PROFILE_ONLY is not None and name not in _PROFILE_ONLY):
        L.check(fn(*args), name)
        return
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    L.check(fn(*args), name)
    b.record()
    _PROFILE.setdefault(name, []).append((a, b))


def _check_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("pagnerf_amd ops need GPU tensors (no CPU fallback); got a tensor on %s" % t.device)


# ------------------------------------------------------------------------------------------- encode
class _EncodeSpec:
    """Static description of a grid encoder (host-side per-level parameters).  Holds plain Python numbers (picklable:
    `torch.save(pipeline)` is the reference's default checkpoint format, config_parser.py:753-756); the ctypes float arrays
    the C ABI takes are built from them once per process."""

    def __init__(self, kind, n_levels, n_feat, **kw):
        self.kind = kind
        self.L = int(n_levels)
        self.F = int(n_feat)
        if kind == "hash":
            self.kw = dict(log2_T=int(kw["log2_T"]), resolutions=_float_list(kw["resolutions"]), half_coords=bool(kw.get("half_coords", False)))
        elif kind == "permuto":
            self.kw = dict(capacity=int(kw["capacity"]), scale_factor=_float_list(kw["scale_factor"]), shift=_float_list(kw["shift"]),
                           half_coords=bool(kw.get("half_coords", False)))
        else:
            raise ValueError(kind)
        self._bind()

    def _bind(self):
        kw = self.kw
        if self.kind == "hash":
            self.log2_T = kw["log2_T"]
            self.res = L.host_floats(kw["resolutions"])
            self.flags = L.ENC_HALF_COORDS if kw["half_coords"] else 0
        else:
            self.capacity = kw["capacity"]
            self.sf = L.host_floats(kw["scale_factor"])
            self.shift = L.host_floats(kw["shift"])
            self.flags = L.ENC_HALF_COORDS if kw["half_coords"] else 0

    def __getstate__(self):
        return dict(kind=self.kind, L=self.L, F=self.F, kw=self.kw)

    def __setstate__(self, state):
        self.kind, self.L, self.F, self.kw = state["kind"], state["L"], state["F"], state["kw"]
        self._bind()

    def rows(self):
        return (1 << self.log2_T) if self.kind == "hash" else self.capacity


def _float_list(values):
    if isinstance(values, torch.Tensor):
        return values.detach().cpu().float().reshape(-1).tolist()
    import numpy as np
    return np.asarray(values, dtype=np.float32).reshape(-1).tolist()


def hash_spec(resolutions, log2_T, n_feat, half_coords=False):
    return _EncodeSpec("hash", len(resolutions), n_feat, log2_T=log2_T, resolutions=resolutions, half_coords=half_coords)


def permuto_spec(scale_factor, shift, capacity, n_feat, half_coords=False):
    """scale_factor, shift: [L,3] (torch CPU / numpy).  half_coords: the kernels round xyz to fp16 first - what the reference's
    `custom_fwd(cast_inputs=torch.half)` does to the coordinates under the trainer's autocast (grids/permuto_grid.py:65,71)."""
    return _EncodeSpec("permuto", len(scale_factor), n_feat, capacity=capacity, scale_factor=scale_factor, shift=shift,
                       half_coords=half_coords)


def _layout_args(t):
    """(stride_m, stride_c, layout) of a feature tensor: [M, C] strided or bf16 [8, M, 8] XCD-grouped."""
    if t.dim() == 3:
        assert t.shape[0] == 8 and t.shape[2] == 8 and t.dtype == torch.bfloat16 and t.is_contiguous()
        return 0, 0, L.LAYOUT_XCD8
    return t.stride(0), t.stride(1), L.LAYOUT_STRIDED


def _encode_fwd(spec, xyz, tables, feat_scale, out, addend=None, flags=None):
    M = xyz.shape[0]
    flags = spec.flags if flags is None else flags
    fs = L.host_floats(feat_scale)
    sm, sc, lay = _layout_args(out)
    if addend is not None:                    # out = bf16(addend + bf16(features)), XCD8 layout only
        assert lay == L.LAYOUT_XCD8 and addend.shape == out.shape and addend.dtype == torch.bfloat16 and addend.is_contiguous()
        if spec.kind == "hash":
            _call("pag_hash_encode_fwd_add", L.ptr(xyz), M, L.ptr(tables), L.dtype_code(tables), spec.L, spec.F, spec.log2_T, spec.res, fs,
                  L.ptr(addend), out.data_ptr(), flags, L.stream())
        else:
            _call("pag_permuto_encode_fwd_add", L.ptr(xyz), M, L.ptr(tables), L.dtype_code(tables), spec.L, spec.F, spec.capacity, spec.sf,
                  spec.shift, fs, L.ptr(addend), out.data_ptr(), flags, L.stream())
        return
    if spec.kind == "hash":
        _call("pag_hash_encode_fwd", L.ptr(xyz), M, L.ptr(tables), L.dtype_code(tables), spec.L, spec.F, spec.log2_T,
              spec.res, fs, out.data_ptr(), L.dtype_code(out), sm, sc, lay, flags, L.stream())
    else:
        _call("pag_permuto_encode_fwd", L.ptr(xyz), M, L.ptr(tables), L.dtype_code(tables), spec.L, spec.F, spec.capacity,
              spec.sf, spec.shift, fs, out.data_ptr(), L.dtype_code(out), sm, sc, lay, flags, L.stream())


_DTYPE_CODE = {torch.float32: L.F32, torch.float16: L.F16, torch.bfloat16: L.BF16}

BWD_ALGO = "binned"     # "binned": atomic-free two-pass scatter (default); "atomic": per-vertex fp32 global atomics


def _encode_bwd(spec, xyz, grad_out, feat_scale, grad_tables, overwrite=False, flags=None):
    """overwrite: grad_tables is uninitialised memory that the binned reduce pass fills completely (pag_*_encode_bwd_set)."""
    lib = L.load()
    flags = spec.flags if flags is None else flags
    M = xyz.shape[0]
    fs = L.host_floats(feat_scale)
    sm, sc, lay = _layout_args(grad_out)
    ws, ws_ptr, ws_bytes = None, None, 0
    if BWD_ALGO == "binned" or lay == L.LAYOUT_XCD8:
        ws_bytes = lib.pag_encode_bwd_workspace_bytes(M, spec.L, spec.F, 8 if spec.kind == "hash" else 4, spec.rows())
        ws = torch.empty(ws_bytes, device=xyz.device, dtype=torch.uint8)
        ws_ptr = ws.data_ptr()
    assert not overwrite or ws_ptr is not None
    suffix = "_set" if overwrite else ""
    if spec.kind == "hash":
        _call("pag_hash_encode_bwd" + suffix, L.ptr(xyz), M, grad_out.data_ptr(), L.dtype_code(grad_out), sm, sc, lay, spec.L, spec.F,
              spec.log2_T, spec.res, fs, L.ptr(grad_tables), ws_ptr, ws_bytes, flags, L.stream())
    else:
        _call("pag_permuto_encode_bwd" + suffix, L.ptr(xyz), M, grad_out.data_ptr(), L.dtype_code(grad_out), sm, sc, lay, spec.L, spec.F,
              spec.capacity, spec.sf, spec.shift, fs, L.ptr(grad_tables), ws_ptr, ws_bytes, flags, L.stream())


def _encode_bwd_xyz(spec, xyz, tables, grad_out, feat_scale, flags=None):
    """d loss / d xyz f32 [M,3] (pose optimisation: ba_pipeline.py:85-92)."""
    M = xyz.shape[0]
    flags = spec.flags if flags is None else flags
    fs = L.host_floats(feat_scale)
    sm, sc, lay = _layout_args(grad_out)
    d_xyz, ws = torch.empty(M, 3, device=xyz.device), torch.empty(8 * M * 3, device=xyz.device)
    if spec.kind == "hash":
        _call("pag_hash_encode_bwd_xyz", L.ptr(xyz), M, L.ptr(tables), L.dtype_code(tables), grad_out.data_ptr(), L.dtype_code(grad_out),
              sm, sc, lay, spec.L, spec.F, spec.log2_T, spec.res, fs, L.ptr(d_xyz), L.ptr(ws), ws.numel() * 4, flags, L.stream())
    else:
        _call("pag_permuto_encode_bwd_xyz", L.ptr(xyz), M, L.ptr(tables), L.dtype_code(tables), grad_out.data_ptr(),
              L.dtype_code(grad_out), sm, sc, lay, spec.L, spec.F, spec.capacity, spec.sf, spec.shift, fs, L.ptr(d_xyz), L.ptr(ws),
              ws.numel() * 4, flags, L.stream())
    return d_xyz


class _Encode(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, tables, spec, feat_scale, out_dtype, feature_major, addend=None, flags=None):
        _check_gpu(xyz, tables)
        ctx.flags = flags
        xyz = xyz.detach().contiguous().float()
        M, C = xyz.shape[0], spec.L * spec.F
        if tables.shape != (spec.L, spec.rows(), spec.F):
            raise RuntimeError("tables shape %s does not match the encoder spec %s" % (tuple(tables.shape), (spec.L, spec.rows(), spec.F)))
        if feature_major == "xcd8":
            out = torch.empty(8, M, 8, device=xyz.device, dtype=torch.bfloat16)
        elif feature_major:
            out = torch.empty(C, M, device=xyz.device, dtype=out_dtype).t()
        else:
            out = torch.empty(M, C, device=xyz.device, dtype=out_dtype)
        tc = tables.detach().contiguous()
        if M:
            _encode_fwd(spec, xyz, tc, feat_scale, out, addend.detach() if addend is not None else None, flags=flags)
        ctx.spec, ctx.feat_scale = spec, feat_scale
        if ctx.needs_input_grad[0]:
            ctx.save_for_backward(xyz, tc)           # d/d xyz needs the table rows again
        else:
            ctx.save_for_backward(xyz)
        ctx.tshape, ctx.tdtype = tables.shape, tables.dtype
        return out

    @staticmethod
    def backward(ctx, g):
        xyz = ctx.saved_tensors[0]
        need_t, need_x = ctx.needs_input_grad[1], ctx.needs_input_grad[0]
        gt = d_xyz = None
        if g.dtype not in (torch.float32, torch.bfloat16):
            g = g.float()
        if g.dim() == 3:
            g = g.contiguous()
        if need_t:
            binned = BWD_ALGO == "binned" or g.dim() == 3
            if xyz.shape[0] and binned:       # the reduce pass writes every row: no zero fill, no read-modify-write
                gt = torch.empty(ctx.tshape, device=xyz.device, dtype=torch.float32)
                _encode_bwd(ctx.spec, xyz, g, ctx.feat_scale, gt, overwrite=True, flags=ctx.flags)
            else:
                gt = torch.zeros(ctx.tshape, device=xyz.device, dtype=torch.float32)
                if xyz.shape[0]:
                    _encode_bwd(ctx.spec, xyz, g, ctx.feat_scale, gt, flags=ctx.flags)
            gt = gt.to(ctx.tdtype)
        if need_x:
            d_xyz = _encode_bwd_xyz(ctx.spec, xyz, ctx.saved_tensors[1], g, ctx.feat_scale, flags=ctx.flags) if xyz.shape[0] \
                else torch.zeros_like(xyz)
        return d_xyz, gt, None, None, None, None, None, None


RAYS_FUSED = os.environ.get("PAG_RAYS_FUSED", "1") != "0"      # A/B switch: position gradient reduced per ray inside the gather pass (_EncodeRays)


class _EncodeRays(torch.autograd.Function):
    """encode() on samples that come straight from the ray march with a pose gradient attached (ray_samples(): samples = origins[ray] +
    dirs[ray] * depth, pc_nerf/ba_pipeline.py:85-92) as ONE autograd node from (origins, dirs, tables) to the features: what the backward wants of
    d loss / d xyz is its per-ray sum and its per-ray sum weighted by depth, and pag_*_encode_bwd_rays forms both inside the gather pass (6 floats per
    wave and ray leave the kernel instead of 8 x 3 floats per sample - 1.4 GB and two launches less on a dense 24 576-ray step).  The samples
    themselves enter detached: their _RaySamples node receives nothing from here."""

    @staticmethod
    def forward(ctx, origins, dirs, xyz, tables, spec, feat_scale, out_dtype, xcd8, flags, depths, pack_start, ridx):
        _check_gpu(xyz, tables, depths, pack_start, ridx)
        xyz = xyz.detach().contiguous().float()
        M, C = xyz.shape[0], spec.L * spec.F
        if tables.shape != (spec.L, spec.rows(), spec.F):
            raise RuntimeError("tables shape %s does not match the encoder spec %s" % (tuple(tables.shape), (spec.L, spec.rows(), spec.F)))
        out = torch.empty(8, M, 8, device=xyz.device, dtype=torch.bfloat16) if xcd8 else torch.empty(M, C, device=xyz.device, dtype=out_dtype)
        tc = tables.detach().contiguous()
        if M:
            _encode_fwd(spec, xyz, tc, feat_scale, out, None, flags=flags)
        ctx.spec, ctx.feat_scale, ctx.flags, ctx.N = spec, feat_scale, flags, origins.shape[0]
        ctx.save_for_backward(xyz, tc, depths.detach().reshape(-1).float().contiguous(), pack_start, ridx.detach().reshape(-1).contiguous())
        ctx.tshape, ctx.tdtype = tables.shape, tables.dtype
        return out

    @staticmethod
    def backward(ctx, g):
        xyz, tc, depths, pack_start, ridx = ctx.saved_tensors
        spec, N, M, dev = ctx.spec, ctx.N, ctx.saved_tensors[0].shape[0], ctx.saved_tensors[0].device
        if g.dtype not in (torch.float32, torch.bfloat16):
            g = g.float()
        g = g.contiguous()
        gt = d_o = d_d = None
        if ctx.needs_input_grad[3]:
            if M and (BWD_ALGO == "binned" or g.dim() == 3):
                gt = torch.empty(ctx.tshape, device=dev, dtype=torch.float32)
                _encode_bwd(spec, xyz, g, ctx.feat_scale, gt, overwrite=True, flags=ctx.flags)
            else:
                gt = torch.zeros(ctx.tshape, device=dev, dtype=torch.float32)
                if M:
                    _encode_bwd(spec, xyz, g, ctx.feat_scale, gt, flags=ctx.flags)
            gt = gt.to(ctx.tdtype)
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            out = torch.empty(N, 6, device=dev)
            lib = L.load()
            ws_bytes = lib.pag_encode_bwd_rays_workspace_bytes(max(M, 1), N)
            ws = torch.empty(ws_bytes // 4, device=dev)
            flags = spec.flags if ctx.flags is None else ctx.flags
            fs = L.host_floats(ctx.feat_scale)
            sm, sc, lay = _layout_args(g)
            tail = (L.ptr(ridx), L.ptr(depths), L.ptr(pack_start), N, L.ptr(out), L.ptr(ws), ws_bytes, flags, L.stream())
            if spec.kind == "hash":
                _call("pag_hash_encode_bwd_rays", L.ptr(xyz), M, L.ptr(tc), L.dtype_code(tc), g.data_ptr(), L.dtype_code(g), sm, sc, lay, spec.L, spec.F, spec.log2_T,
                      spec.res, fs, *tail)
            else:
                _call("pag_permuto_encode_bwd_rays", L.ptr(xyz), M, L.ptr(tc), L.dtype_code(tc), g.data_ptr(), L.dtype_code(g), sm, sc, lay, spec.L, spec.F, spec.capacity,
                      spec.sf, spec.shift, fs, *tail)
            d_o, d_d = out[:, :3], out[:, 3:]
        return d_o, d_d, None, gt, None, None, None, None, None, None, None, None


def encode(xyz, tables, spec, feat_scale=None, out_dtype=torch.float32, feature_major=False, layout=None, addend=None, half_coords=None, rays=None):
    """Grid features [M, L*F] (column = level*F + f); layout="xcd8" returns the bf16 [8, M, 8] XCD-grouped tensor the
    fused decoders consume directly.  Differentiable w.r.t. tables and (when xyz.requires_grad) xyz.
    addend (xcd8 only, treated as a constant): the result is bf16(addend + bf16(features)) in the same launch.
    half_coords: None = as the spec says; True / False override it for this call (forward and both gradients).
    rays = ray_samples()'s tag of xyz (origins, dirs, depths, pack_start, ridx): the position gradient then goes to origins / dirs directly (_EncodeRays)."""
    flags = None if half_coords is None else (L.ENC_HALF_COORDS if half_coords else 0)
    if rays is not None and RAYS_FUSED and addend is None and not feature_major and xyz.requires_grad and torch.is_grad_enabled() and xyz.shape[0]:
        origins, dirs, depths, pack_start, ridx = rays
        if depths.numel() == xyz.shape[0] and ridx.numel() == xyz.shape[0]:
            # (xyz enters detached: as a differentiable input its _RaySamples node would still be run - on materialised zeros - by the engine)
            return _EncodeRays.apply(origins, dirs, xyz.detach(), tables, spec, feat_scale, out_dtype, layout == "xcd8", flags, depths, pack_start, ridx)
    return _apply(_Encode, xyz, tables, spec, feat_scale, out_dtype, "xcd8" if layout == "xcd8" else feature_major, addend, flags)


def xcd8_supported(n_levels, n_feat):
    return ((n_levels + 7) // 8) * n_feat <= 8


def xcd8_level(g, j):
    """Level in slot j of XCD group g (csrc/common.h xcd8_level): even bands of 8 levels ascend with g, odd bands descend."""
    return 8 * j + 7 - g if j & 1 else 8 * j + g


def xcd8_columns(n_levels, n_feat):
    """staged position p = 8*g + e  ->  feature column level*F + f (or -1 for padding), as a list of 64."""
    cols = []
    for p in range(64):
        g, e = p >> 3, p & 7
        j, f = divmod(e, n_feat)
        level = xcd8_level(g, j)
        cols.append(level * n_feat + f if (j < (n_levels + 7) // 8 and level < n_levels) else -1)
    return cols


# ---------------------------------------------------------------------------------------------- MLP
def _mm_f32(a, b):
    """a^T-free helper: a [K,M] @ b [M,N] with fp32 result (bf16 inputs accumulate in fp32)."""
    if a.dtype == torch.float32:
        return a @ b
    try:
        return torch.mm(a, b, out_dtype=torch.float32)
    except (TypeError, RuntimeError):
        return (a @ b).float()


def _recompute_ok(ctx, x1, x2, k1, in_dim, n_layers, out_dim, mode, grouped, stats_only, M, out_act, out_dtype, rank1_expected, col0):
    """Forward-time mirror of csrc/mlp.hip::fused_kind() - the shapes whose backward (pag_mlp_bwd with wgrad_workspace) recomputes the hidden
    activations, so that the forward need not write them.  Every condition fused_kind() checks that is knowable at forward time is
    checked here (batch bound, layout, widths, output activation / dtype per kernel kind, whether the upstream gradient will be dense or
    rank-1); a mismatch that still slips through (e.g. a _ColourDensity backward that receives no sigma gradient) takes
    _recompute_hidden(), which is correct but costs a second forward - it warns once."""
    if not WGRAD_FUSED or mode != L.MLP_MFMA_BF16 or x1.dtype != torch.bfloat16 or not ctx.needs_input_grad[0]:
        return False
    if ctx.needs_input_grad[1] and grouped is not None:       # only the colour-like kernel hands out dz_0 for the per-ray input's gradient
        return False
    if not (0 < M <= L.MLP_FUSED_WIDE_MAX_M) or n_layers not in (2, 3):
        return False
    if grouped is not None:
        levels, feats = grouped
        if levels < 1 or feats < 1 or ((levels + 7) // 8) * feats > 8:
            return False
        j = 7 // feats
        if j < (levels + 7) // 8 and xcd8_level(7, j) < levels:      # staged position 63 is a real feature: no room for the bias column
            return False
        if x2 is not None:
            return False
        if out_dim > 32:                                      # wide softmax head: stage A + stage B (rank-1 gradient, statistics-only forward)
            return stats_only and n_layers == 3 and 192 < out_dim <= 224 and out_act == L.ACT_SOFTMAX and out_dtype == torch.bfloat16
        if rank1_expected:                                    # semantic-like: composited softmax head
            return out_act == L.ACT_SOFTMAX and out_dim <= 8 and out_dtype == torch.bfloat16
        return out_act == L.ACT_NONE and out_dim % 4 == 0 and out_dtype == torch.bfloat16      # density-like: dense bf16 gradient
    return (x2 is not None and k1 == 16 and x2.shape[1] == 32 and in_dim <= 48 and out_dim <= 4 and out_act == L.ACT_SIGMOID
            and out_dtype == torch.float32 and col0 and not rank1_expected)


_WARNED_RECOMPUTE = False


def _recompute_hidden(x1, x2, x2_index, Wc, bc, in_dim, k1, grouped, mode, hidden):
    """Fill the missing entries of `hidden` by re-running the forward (pag_mlp_fwd with hidden_save) - the fallback for a backward
    that the fused kernels cannot serve although the forward did not save the activations."""
    global _WARNED_RECOMPUTE
    if not _WARNED_RECOMPUTE:
        _WARNED_RECOMPUTE = True
        import warnings
        warnings.warn("pagnerf_amd: a decoder backward could not use the fused kernels although its forward skipped the hidden activations; "
                      "re-running the forward for them (correct, but one extra launch and an [M, out] scratch per backward)")
    M = x1.shape[1] if grouped is not None else x1.shape[0]
    n_layers = len(Wc)
    dev = x1.device
    full = [h if h is not None else torch.empty(M, 64, device=dev, dtype=torch.bfloat16 if mode == L.MLP_MFMA_BF16 else torch.float32) for h in hidden]
    out_dim = Wc[-1].shape[0]
    a = L.MlpFwdArgs()
    a.x1, a.x1_dtype, a.k1 = L.ptr(x1), L.dtype_code(x1), k1
    if grouped is not None:
        a.x1_layout, a.x1_levels, a.x1_feats = L.LAYOUT_XCD8, grouped[0], grouped[1]
    if x2 is not None:
        a.x2, a.k2p, a.x2_index = L.ptr(x2), x2.shape[1], L.ptr(x2_index)
    a.in_dim, a.n_layers, a.out_dim = in_dim, n_layers, out_dim
    for i in range(n_layers):
        a.W[i], a.b[i] = L.ptr(Wc[i]), L.ptr(bc[i])
    scratch = torch.empty(M, out_dim, device=dev, dtype=torch.bfloat16)
    a.out_act, a.out, a.out_dtype, a.mode = L.ACT_NONE, L.ptr(scratch), L.BF16, mode
    for i, h in enumerate(full):
        a.hidden_save[i] = L.ptr(h)
    _call("pag_mlp_fwd", ctypes.byref(a), M, L.stream())
    return full


_NO_FAST_FWD = os.environ.get("PAG_NO_FAST_FWD") is not None
_OUTER_GRAD = True          # grad mode of the code that called the decoder op (see _apply_decoder)


class _InferCtx:
    """Stand-in for the autograd ctx of a Function whose result nobody will differentiate (torch.no_grad(): validation renders, prune): nothing is saved,
    nothing needs a gradient; attributes a forward() parks on the ctx for its backward land on a plain object and die with it."""
    saved_tensors = ()

    def __init__(self, n):
        self.needs_input_grad = (False,) * n

    def save_for_backward(self, *tensors):
        pass

    def mark_non_differentiable(self, *tensors):
        pass

    def set_materialize_grads(self, value):
        pass

    def __getattr__(self, name):          # only reached for attributes nobody set: the optional hand-offs (fwd_pair, fwd_hold, ...) read as absent
        if name.startswith("__"):
            raise AttributeError(name)
        return None


def _apply(fn, *args):
    """fn.apply(*args); under torch.no_grad() the forward is called directly with an _InferCtx - torch.autograd.Function.apply costs 10 - 15 us of host time
    per call whether or not a graph is recorded, six to eight of them per forward-only trace (a third of the host side of a validation pack)."""
    if torch.is_grad_enabled():
        return fn.apply(*args)
    return fn.forward(_InferCtx(len(args)), *args)


def _apply_decoder(fn, *args):
    """fn.apply(*args) with the caller's grad mode visible to the forward: under torch.no_grad() (validation renders, prune) the decoder
    launches keep nothing for a backward - no hidden activations, no softmax statistics beyond what the statistics-only wide head hands to
    its own compositing launch."""
    global _OUTER_GRAD
    prev, _OUTER_GRAD = _OUTER_GRAD, torch.is_grad_enabled()
    try:
        return _apply(fn, *args)
    finally:
        _OUTER_GRAD = prev


class _FusedMLP(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x1, x2, x2_index, in_dim, out_act, mode, out_dtype, grouped, *wb):
        _check_gpu(x1, x2, x2_index, *wb)
        n_layers = len(wb) // 2
        Ws, bs = wb[:n_layers], wb[n_layers:]
        if grouped is not None:
            M, k1 = x1.shape[1], 64
        else:
            M, k1 = x1.shape
        x1 = x1.detach()
        if not x1.is_contiguous():
            x1 = x1.contiguous()
        if x1.dtype not in (torch.float32, torch.bfloat16):
            x1 = x1.float()
        out_dim = Ws[-1].shape[0]
        # under torch.no_grad() (validation renders, prune) nothing is kept for a backward: no hidden activations, no softmax statistics
        # beyond what the statistics-only wide head hands to its compositing launch
        # (_OUTER_GRAD: the grad mode of the CALLER, recorded by _apply_decoder - inside an autograd.Function's forward grad mode is always
        # off, and ctx.needs_input_grad / tensor.requires_grad still say True for a parameter inside a no_grad region)
        any_grad = getattr(ctx, "any_grad", None)
        need_grad = _OUTER_GRAD and bool(any(ctx.needs_input_grad) if any_grad is None else any_grad)
        wide_softmax = mode == L.MLP_MFMA_BF16 and out_act == L.ACT_SOFTMAX and out_dim > 64 and out_dtype == torch.bfloat16
        stats_only = bool(getattr(ctx, "stats_only", False)) and wide_softmax       # _HeadComposite: no [M,out] tensor at all
        out = None if stats_only else torch.empty(M, out_dim, device=x1.device, dtype=out_dtype)
        hdt = torch.bfloat16 if mode == L.MLP_MFMA_BF16 else torch.float32
        # The fused backward kernels RECOMPUTE the hidden activations from x1 (bit-identical MFMA sequence), so the forward of a
        # decoder they will serve does not write them: 268 MB per hidden layer at M = 2.1 M, in launches that run at the HBM rate.
        # Wide head: the LAST hidden layer is kept (the probabilities are rebuilt from it).  A backward that turns out not to be
        # fusable (no column-0 gradient, d x2 requested, ...) re-runs the forward for them (_recompute_hidden: rare, slow, correct).
        recompute = need_grad and M > 0 and _recompute_ok(ctx, x1, x2, k1, in_dim, n_layers, out_dim, mode, grouped, stats_only, M, out_act,
                                                          out_dtype, bool(getattr(ctx, "rank1_expected", False)),
                                                          bool(getattr(ctx, "want_col0_relu", False)))
        hidden = []
        if need_grad or stats_only:
            for i in range(n_layers - 1):
                keep = (need_grad and not recompute) or (stats_only and i == n_layers - 2)
                hidden.append(torch.empty(M, 64, device=x1.device, dtype=hdt) if keep else None)
        a = L.MlpFwdArgs()
        a.x1, a.x1_dtype, a.k1 = L.ptr(x1), L.dtype_code(x1), k1
        if grouped is not None:
            a.x1_layout, a.x1_levels, a.x1_feats = L.LAYOUT_XCD8, grouped[0], grouped[1]
        if x2 is not None:
            x2 = x2.detach().contiguous().float()
            x2_index = x2_index.detach().contiguous()
            if x2_index.dtype != torch.int32:
                x2_index = x2_index.int()
            a.x2, a.k2p, a.x2_index = L.ptr(x2), x2.shape[1], L.ptr(x2_index)
        a.in_dim, a.n_layers, a.out_dim = in_dim, n_layers, out_dim
        Wc = [w.detach().contiguous().float() for w in Ws]
        bc = [b.detach().contiguous().float() for b in bs]
        for i in range(n_layers):
            a.W[i], a.b[i] = L.ptr(Wc[i]), L.ptr(bc[i])
        a.out_act, a.out, a.out_dtype, a.mode = out_act, L.ptr(out), _DTYPE_CODE[out_dtype], mode
        for i, h in enumerate(hidden):
            a.hidden_save[i] = L.ptr(h)
        # wide softmax heads: per-sample (max*log2e, 1/sum) lets the backward rebuild the probabilities from the saved
        # hidden layer instead of streaming `out` through twice (pag_mlp_bwd_args.softmax_stats)
        stats = None
        if wide_softmax and (need_grad or stats_only):
            stats = torch.empty(M, 2, device=x1.device)
            a.softmax_stats = L.ptr(stats)
        ctx.col0_relu = None
        if getattr(ctx, "want_col0_relu", False) and mode == L.MLP_MFMA_BF16 and grouped is None and x1.dtype == torch.bfloat16 \
                and out_dim <= 64 and M:
            ctx.col0_relu = torch.empty(M, device=x1.device)           # _ColourDensity: sigma from the same launch
            a.x1_col0_relu = L.ptr(ctx.col0_relu)
        # A pair of decoders on one input (_HeadCompositePair): the narrow one's launch is PREPARED and parked (ctx.fwd_hold) and rides
        # in the wide one's call (ctx.fwd_pair -> pag_mlp_fwd_args.pair) where the library can; else the caller issues it.
        hold, pair_hold = getattr(ctx, "fwd_hold", None), getattr(ctx, "fwd_pair", None)
        global _NEXT_HOLD, _NEXT_PRODUCER
        if hold is None and _NEXT_HOLD is not None and _NEXT_HOLD.get("x1_ptr") == x1.data_ptr():
            # decoder_hold(x1): this launch is prepared and parked for the decoder that consumes its output
            hold, _NEXT_HOLD = _NEXT_HOLD, None
            if not (M and mode == L.MLP_MFMA_BF16 and all(t is None for t in hidden) and stats is None and ctx.col0_relu is None):
                hold = None                              # nothing a consumer's launch could carry: issue it now
        producer, _NEXT_PRODUCER = _NEXT_PRODUCER, None
        if producer is not None and producer.get("args") is not None and not producer.get("taken"):
            # the decoder whose output is this one's x1 waits in `producer`: in this launch where the library can (pag_mlp_fwd_args.x1_producer),
            # else its own launch FIRST
            if M and x1.data_ptr() == producer["out_ptr"] and L.load().pag_mlp_fwd_producer_supported(ctypes.byref(a), ctypes.byref(producer["args"]), M) == 1:
                a.x1_producer = ctypes.pointer(producer["args"])
            else:
                _call("pag_mlp_fwd", ctypes.byref(producer["args"]), producer["M"], L.stream())
            producer["taken"] = True
        if M and hold is not None:
            hold["out_ptr"] = out.data_ptr() if out is not None else 0
            hold["args"], hold["M"] = a, M
            hold["keep"] = [v for v in locals().values() if isinstance(v, (torch.Tensor, list, tuple))]
        elif M:
            if pair_hold is not None and pair_hold.get("args") is not None and M <= L.MLP_FUSED_WIDE_MAX_M and not _NO_FAST_FWD \
                    and L.load().pag_mlp_fwd_pair_supported(ctypes.byref(a), ctypes.byref(pair_hold["args"])) == 1:
                a.pair = ctypes.pointer(pair_hold["args"])
                pair_hold["taken"] = True
            # _HeadComposite: the per-ray weighted sum in the decoder's own launch, one pass over the logits (pag_mlp_fwd_args.composite)
            comp = getattr(ctx, "fwd_composite", None)
            ctx.composited = False
            if comp is not None and HEAD_FWD_ONCE and stats_only and L.load().pag_mlp_fwd_composite_supported(ctypes.byref(a), M) == 1:
                hc = L.HeadCompositeArgs()
                hc.pack_start, hc.ray_of_pack, hc.P = L.ptr(comp["pack_start"]), L.ptr(comp["ray_of_pack"]), comp["ray_of_pack"].shape[0]
                hc.weights, hc.alpha, hc.out = L.ptr(comp["weights"]), L.ptr(comp["alpha"]), L.ptr(comp["out"])
                hc.n_samples = int(M if SAMPLES_HINT is None else SAMPLES_HINT)
                a.composite = ctypes.pointer(hc)
                ctx.composited = True
            _call("pag_mlp_fwd", ctypes.byref(a), M, L.stream())
        ctx.cfg = (in_dim, out_act, mode, n_layers, k1, grouped)
        ctx.x2_packs = None
        ctx.has_stats = stats is not None
        ctx.out_dtype = out_dtype
        extra = (stats, bc[-1]) if stats is not None else ()
        ctx.save_for_backward(x1, x2, x2_index, out, *hidden, *Wc, *extra)
        ctx.n_hidden = len(hidden)
        ctx.bc = bc
        if stats_only:
            ctx.fwd_state = (hidden[-1], Wc[-1], bc[-1], stats)
        return out

    @staticmethod
    def backward(ctx, g):
        return _FusedMLP._backward_impl(ctx, g, None)

    @staticmethod
    def _backward_impl(ctx, g, rank1, dx1_into=None, col0_add=None, col0_gate=None, wgrad_queue=None, dx1_out=None, hold=None,
                       pair_hold=None):
        """rank1 = (g_ray f32 [N,out], g_scale f32 [M], g_index i32 [M]) replaces the dense upstream gradient g.
        dx1_into: an XCD8 gradient tensor of another decoder on the same input - this one's d x1 is added to it in place.
        dx1_out: write d x1 into this tensor instead of a new one.
        hold (dict): a fused launch is PREPARED, not issued - args and everything they point to are parked in the dict, to ride in the
        call of the other decoder of a pair (pair_hold = that dict; pag_mlp_bwd_args.pair) or to be issued by the caller."""
        lib = L.load()
        in_dim, out_act, mode, n_layers, k1, grouped = ctx.cfg
        saved = ctx.saved_tensors
        x1, x2, x2_index, out = saved[:4]
        hidden = list(saved[4:4 + ctx.n_hidden])
        Wc = list(saved[4 + ctx.n_hidden:4 + ctx.n_hidden + n_layers])
        stats, b_last = (saved[-2], saved[-1]) if getattr(ctx, "has_stats", False) else (None, None)
        M = x1.shape[1] if grouped is not None else x1.shape[0]
        out_dim = Wc[-1].shape[0]
        dev = x1.device
        zdt = torch.bfloat16 if mode == L.MLP_MFMA_BF16 else torch.float32
        out_dtype = ctx.out_dtype
        need_dx = ctx.needs_input_grad[0]
        need_dx2 = ctx.needs_input_grad[1] and x2 is not None
        dx1 = (dx1_into if dx1_into is not None else (dx1_out if dx1_out is not None else torch.empty(x1.shape, device=dev, dtype=x1.dtype))) \
            if need_dx else None
        k2p = x2.shape[1] if x2 is not None else 0
        a = L.MlpBwdArgs()
        if rank1 is None:
            g = g.contiguous().to(out_dtype)       # grad_out travels in the output's dtype
            a.grad_out = L.ptr(g)
        else:
            g_ray, g_scale, g_index = rank1[:3]
            a.g_ray, a.g_scale, a.g_index = L.ptr(g_ray), L.ptr(g_scale), L.ptr(g_index)
            if len(rank1) > 3:
                g_ray_scale = rank1[3].float()            # named: stays alive past the launch
                a.g_ray_scale = L.ptr(g_ray_scale)
        a.out, a.out_dtype, a.out_act = L.ptr(out), _DTYPE_CODE[out_dtype], out_act
        a.k1, a.in_dim, a.n_layers, a.out_dim = k1, in_dim, n_layers, out_dim
        if grouped is not None:
            a.x1_layout, a.x1_levels, a.x1_feats = L.LAYOUT_XCD8, grouped[0], grouped[1]
        for i in range(n_layers):
            a.W[i] = L.ptr(Wc[i])
        bc = getattr(ctx, "bc", None)
        if bc is not None:
            for i in range(n_layers):
                a.b[i] = L.ptr(bc[i])
        a.dx1, a.dx1_dtype, a.mode = L.ptr(dx1), (L.dtype_code(dx1) if need_dx else 0), mode
        a.softmax_stats, a.b_last = L.ptr(stats), L.ptr(b_last)
        a.dx1_accumulate = 1 if (need_dx and dx1_into is not None) else 0
        fuse_col0 = col0_add is not None and need_dx and grouped is None and mode == L.MLP_MFMA_BF16 and out_dim <= 64
        if col0_gate is not None and not fuse_col0:          # the gate only exists on the fused path's preconditions
            col0_add, col0_gate = col0_add * (col0_gate > 0), None
        if fuse_col0:
            a.dx1_col0_add = L.ptr(col0_add)
            if col0_gate is not None:
                a.dx1_col0_gate = L.ptr(col0_gate)
        # Weight gradients inside the backward-data launch (pag_mlp_bwd_args.wgrad_workspace) where the library has a fused
        # kernel for this decoder shape: no dz tensors, no second pass over the activations.  d x2 (pose optimisation) is formed
        # from dz_0, so that case keeps the dz tensors and the separate weight-gradient launches.
        fused = False
        dz0_fused = dz0_slots = None
        if WGRAD_FUSED and 0 < M <= L.MLP_FUSED_WIDE_MAX_M and mode == L.MLP_MFMA_BF16 and x1.dtype == torch.bfloat16 \
                and (not need_dx2 or grouped is None):
            a.x1, a.x1_dtype = L.ptr(x1), L.BF16
            if x2 is not None:
                a.x2, a.k2p, a.x2_index = L.ptr(x2), k2p, L.ptr(x2_index)
            fused = lib.pag_mlp_bwd_fused_supported(ctypes.byref(a)) == 1
            if fused and need_dx2:        # colour-like kernel: d x2 is formed from dz_0 below
                packs = ctx.x2_packs
                if DZ0_SLOTS and packs is not None and _one_pack_per_ray(packs[1], x2.shape[0]) and packs[0].shape[0] == x2.shape[0] + 1:
                    # samples packed ray by ray (x2_index non-decreasing): the per-ray sum of dz_0 starts inside the kernel - one 64-float row per
                    # (tile, ray) instead of the [M,64] tensor (pag_mlp_bwd_args.dz0_slots)
                    dz0_slots = torch.empty(lib.pag_mlp_dz0_slots_bytes(M, x2.shape[0]) // 4, device=dev)
                    a.dz0_slots = L.ptr(dz0_slots)
                else:                     # dz_0 comes out as a tensor (pag_mlp_bwd_args.dz[0])
                    dz0_fused = torch.empty(M, 64, device=dev, dtype=torch.bfloat16)
                    a.dz[0] = L.ptr(dz0_fused)
        if not fused and any(h is None for h in hidden):      # the forward counted on the fused kernels: rebuild what it skipped
            hidden = _recompute_hidden(x1, x2, x2_index, Wc, bc, in_dim, k1, grouped, mode, hidden)
        for i, h in enumerate(hidden):
            a.hidden_save[i] = L.ptr(h)
        dz = None
        gW, gb = [], []
        if fused:
            ws_bytes = lib.pag_mlp_bwd_fused_workspace_bytes(ctypes.byref(a), M)
            ws = torch.empty(ws_bytes // 4, device=dev)
            a.wgrad_workspace, a.wgrad_workspace_bytes = L.ptr(ws), ws_bytes
            for l in range(n_layers):
                gW.append(torch.empty(Wc[l].shape[0], in_dim if l == 0 else 64, device=dev))
                gb.append(torch.empty(Wc[l].shape[0], device=dev))
                a.dW[l], a.db[l] = L.ptr(gW[l]), L.ptr(gb[l])
        else:
            dz = [torch.empty(M, 64, device=dev, dtype=zdt) for _ in range(n_layers - 1)] + [torch.empty(M, out_dim, device=dev, dtype=zdt)]
            for i in range(n_layers):
                a.dz[i] = L.ptr(dz[i])
        if M and hold is not None:
            hold["args"], hold["M"], hold["fused"] = a, M, fused       # issued later: inside the pair's other call or by the caller
            hold["keep"] = [v for v in locals().values() if isinstance(v, (torch.Tensor, list, tuple))]
        elif M:
            if pair_hold is not None and fused and pair_hold.get("fused") \
                    and lib.pag_mlp_bwd_pair_supported(ctypes.byref(a), ctypes.byref(pair_hold["args"])) == 1:
                a.pair = ctypes.pointer(pair_hold["args"])
                pair_hold["taken"] = True
            _call("pag_mlp_bwd", ctypes.byref(a), M, L.stream())
        if col0_add is not None and need_dx and not fuse_col0:
            dx1[:, 0] += col0_add.to(dx1.dtype)          # parity path: same sum with torch ops
        if fused:
            pass
        elif mode == L.MLP_MFMA_BF16 and M:
            # weight gradients on the matrix cores: per-workgroup fp32 slabs, summed here (deterministic)
            specs = []
            for l in range(n_layers):
                n_out = Wc[l].shape[0]
                sp = dict(dz=dz[l], n_out=n_out, a2=None, k2p=0, a2_index=None, levels=0, feats=0)
                if l == 0 and grouped is not None:
                    sp.update(a1=x1, a1_dtype=L.BF16, a1_layout=L.LAYOUT_XCD8, k1=64, n_in=64, levels=grouped[0], feats=grouped[1])
                elif l == 0:
                    sp.update(a1=x1, a1_dtype=L.dtype_code(x1), a1_layout=L.LAYOUT_STRIDED, k1=k1, n_in=in_dim,
                              a2=x2, k2p=(x2.shape[1] if x2 is not None else 0), a2_index=x2_index)
                else:
                    sp.update(a1=hidden[l - 1], a1_dtype=L.BF16, a1_layout=L.LAYOUT_STRIDED, k1=64, n_in=64)
                sp["w"] = torch.empty(n_out, in_dim if l == 0 else 64, device=dev)
                sp["b"] = torch.empty(n_out, device=dev)
                gW.append(sp["w"])
                gb.append(sp["b"])
                specs.append(sp)
            if wgrad_queue is not None:
                wgrad_queue.extend(specs)          # the caller launches several decoders' weight gradients together
            else:
                _launch_wgrad(specs, M)
        else:
            # fp32 parity path: dz_l^T @ input_l as plain fp32 GEMMs (BLAS)
            for l in range(n_layers):
                z = dz[l]
                if l == 0:
                    n1 = min(k1, in_dim)
                    w = _mm_f32(z.t(), x1[:, :n1].to(z.dtype))
                    if in_dim > k1:
                        seg = torch.zeros(x2.shape[0], 64, device=dev, dtype=torch.float32)
                        seg.index_add_(0, x2_index.long(), z.float())
                        w = torch.cat([w, seg.t() @ x2[:, :in_dim - k1]], dim=1)
                else:
                    w = _mm_f32(z.t(), hidden[l - 1])
                gW.append(w)
                gb.append(z.sum(0, dtype=torch.float32))
        dx2 = None
        if need_dx2 and dz0_fused is not None:
            dz = [dz0_fused]
        if need_dx2:
            # d x2[r] = (sum of dz_0 over the samples that gathered row r) @ W_0[:, k1:]  - the per-ray view embedding's
            # gradient (pose optimisation: the view direction depends on the camera rotation, ba_pipeline.py:89-90)
            R = x2.shape[0]
            if dz0_slots is not None and M:                # the kernel left per-(tile, ray) sums: add each ray's rows
                seg = torch.empty(R, 64, device=dev)
                _call("pag_mlp_dz0_slots_sum", L.ptr(ctx.x2_packs[0]), R, L.ptr(dz0_slots), L.ptr(seg), L.stream())
            elif M == 0:
                seg = torch.zeros(R, 64, device=dev)
            elif ctx.x2_packs is not None:                 # rows gathered pack by pack: one segmented-sum launch
                pack_start, ray_of_pack = ctx.x2_packs
                seg = composite_feats(dz[0], torch.ones(M, device=dev), torch.ones(R, device=dev), pack_start, ray_of_pack, R)
            else:
                seg = torch.zeros(R, dz[0].shape[1], device=dev).index_add_(0, x2_index.long(), dz[0].float())
            w_tail = Wc[0][:, k1:in_dim]
            dx2 = torch.zeros_like(x2)
            dx2[:, :in_dim - k1] = seg @ w_tail
        return (dx1, dx2, None, None, None, None, None, None, *gW, *gb)


DZ0_SLOTS = os.environ.get("PAG_DZ0_SLOTS", "1") != "0"      # A/B switch: per-(tile, ray) sums of dz_0 inside the colour backward (pag_mlp_bwd_args.dz0_slots)
WGRAD_MAX_BATCH = 6      # WG_MAX_BATCH of csrc/mlp.hip
WGRAD_FUSED = os.environ.get("PAG_NO_FUSED_WGRAD") is None      # narrow decoders: weight gradients inside pag_mlp_bwd (no dz tensors)


def _launch_wgrad(specs, M):
    """Weight / bias gradients of the given decoder layers (dicts built by _FusedMLP._backward_impl; all over the same M samples)
    through pag_mlp_wgrad_batch: the narrow layers share ONE slab launch (grid.y = layer) whose ~1024 workgroups - one wave of
    workgroups on the chip; a second, partial wave cost 10 % - are split between them; one finish launch sums every layer's slabs."""
    lib = L.load()
    nblk_max = lib.pag_mlp_wgrad_blocks(M)
    for c0 in range(0, len(specs), WGRAD_MAX_BATCH):
        part = specs[c0:c0 + WGRAD_MAX_BATCH]
        n_narrow = max(1, sum(1 for sp in part if sp["n_out"] <= 64))
        layers = (L.WgradLayer * len(part))()
        keep = []
        for y, sp in zip(layers, part):
            n_out = sp["n_out"]
            nblk = max(1, nblk_max // n_narrow) if n_out <= 64 else min(nblk_max, 512)      # wide layers: fewer, larger slabs
            slabs = torch.empty(nblk, (n_out + 31) // 32 * 32, 96, device=sp["dz"].device)
            keep.append(slabs)
            y.dz, y.dz_cols, y.n_out = L.ptr(sp["dz"]), sp["dz"].shape[1], n_out
            y.a1, y.a1_dtype, y.a1_layout, y.k1, y.n_in = L.ptr(sp["a1"]), sp["a1_dtype"], sp["a1_layout"], sp["k1"], sp["n_in"]
            y.a2, y.k2p, y.a2_index = L.ptr(sp["a2"]), sp["k2p"], L.ptr(sp["a2_index"])
            y.a1_levels, y.a1_feats = sp["levels"], sp["feats"]
            y.slabs, y.n_blocks, y.dW, y.db = L.ptr(slabs), nblk, L.ptr(sp["w"]), L.ptr(sp["b"])
        _call("pag_mlp_wgrad_batch", layers, len(part), M, L.stream())


class _AffineXCD8(torch.autograd.Function):
    """out f32 [M, n_out] = x . W^T + b on the encoders' bf16 [8, M, 8] features (pag_affine_xcd8_fwd); backward: d x in the same
    layout (pag_affine_xcd8_bwd_dx), d W / d b through the decoders' weight-gradient kernels (pag_mlp_wgrad_batch)."""

    @staticmethod
    def forward(ctx, x8, W, b, grouped):
        _check_gpu(x8, W, b)
        M, (n_out, in_dim) = x8.shape[1], W.shape
        Wc, bc = W.detach().float().contiguous(), b.detach().float().contiguous()
        out = torch.empty(M, n_out, device=x8.device)
        _call("pag_affine_xcd8_fwd", L.ptr(x8), M, grouped[0], grouped[1], L.ptr(Wc), L.ptr(bc), n_out, in_dim, L.ptr(out), L.stream())
        ctx.save_for_backward(x8, Wc)
        ctx.grouped = grouped
        return out

    @staticmethod
    def backward(ctx, g):
        x8, Wc = ctx.saved_tensors
        M, (n_out, in_dim) = x8.shape[1], Wc.shape
        g = g.float().contiguous()
        dx = dW = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x8)
            _call("pag_affine_xcd8_bwd_dx", L.ptr(g), M, ctx.grouped[0], ctx.grouped[1], L.ptr(Wc), n_out, in_dim, L.ptr(dx), L.stream())
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dW, db = torch.empty(n_out, in_dim, device=g.device), torch.empty(n_out, device=g.device)
            if M:
                _launch_wgrad([dict(dz=g.to(torch.bfloat16), n_out=n_out, a1=x8, a1_dtype=L.BF16, a1_layout=L.LAYOUT_XCD8, k1=64, n_in=64,
                                    a2=None, k2p=0, a2_index=None, levels=ctx.grouped[0], feats=ctx.grouped[1], w=dW, b=db)], M)
            else:
                dW.zero_(), db.zero_()
        return dx, dW, db, None


def affine_xcd8(x8, W, b, grouped):
    """x8 bf16 [8, M, 8] (grouped = (levels, feats)), W [n_out, levels*feats], b [n_out] -> f32 [M, n_out]."""
    return _AffineXCD8.apply(x8, W, b, grouped)


class _ColourDensity(_FusedMLP):
    """The colour decoder on cat(density_feats, PE) AND the density sigma = relu(density_feats[:, 0]) that
    pc_nerf/panoptic_delta_nef.py:188 reads off its x1, as one autograd node: the gradient of sigma is added to column 0 of
    the decoder's d x1 inside the backward kernel - no zero-padded [M,16] gradient, slice / cast / relu backward or add pass."""

    @staticmethod
    def forward(ctx, x1, x2, x2_index, in_dim, out_act, mode, out_dtype, grouped, *wb):
        ctx.want_col0_relu = True
        rgb = _FusedMLP.forward(ctx, x1, x2, x2_index, in_dim, out_act, mode, out_dtype, grouped, *wb)
        if ctx.col0_relu is not None:            # written by the decoder launch; the backward gates on it inside the kernel
            ctx.pre = None
            return rgb, ctx.col0_relu
        pre = x1.detach()[:, 0].float()
        ctx.pre = pre
        return rgb, torch.relu(pre)

    @staticmethod
    def backward(ctx, g_rgb, g_sigma):
        saved_out = ctx.saved_tensors[3]
        if g_rgb is None:
            g_rgb = torch.zeros_like(saved_out)
        if g_sigma is None:
            return _FusedMLP._backward_impl(ctx, g_rgb, None)
        if ctx.pre is None:
            return _FusedMLP._backward_impl(ctx, g_rgb, None, col0_add=g_sigma.float().contiguous(), col0_gate=ctx.col0_relu)
        add = (g_sigma.float() * (ctx.pre > 0)).contiguous()
        return _FusedMLP._backward_impl(ctx, g_rgb, None, col0_add=add)


def colour_and_density(x1, weights, biases, x2, x2_index, in_dim, out_act=L.ACT_SIGMOID, mode=L.MLP_MFMA_BF16,
                       out_dtype=torch.float32, x2_packs=None, producer=None):
    """-> (rgb [M,3], sigma f32 [M] = relu(x1[:,0])); x1 = the density decoder's [M,16] output (see _ColourDensity).
    producer: the decoder_hold() in which the density decoder's launch waits - evaluated in this decoder's launch where the library can."""
    global _NEXT_PRODUCER
    _NEXT_PRODUCER = producer
    try:
        rgb, sigma = _apply_decoder(_ColourDensity, x1, x2, x2_index, int(in_dim), out_act, mode, out_dtype, None, *weights, *biases)
    finally:
        _NEXT_PRODUCER = None            # a call that raised before _FusedMLP.forward took the producer must not leave it for an unrelated decoder
    if x2_packs is not None and x2 is not None and x2.requires_grad and rgb.grad_fn is not None:
        rgb.grad_fn.x2_packs = x2_packs
    return rgb, sigma


def fused_mlp(x1, weights, biases, x2=None, x2_index=None, in_dim=None, out_act=L.ACT_NONE, mode=L.MLP_MFMA_BF16,
              out_dtype=torch.float32, x1_grouped=None, x2_packs=None):
    """wisp BasicDecoder (Linear+ReLU ... Linear) [+ sigmoid/softmax] in one launch.
    x1 [M,k1] (+ optional per-ray x2 [R,k2p] gathered by x2_index [M]); weights[i] is [out,in].
    x1_grouped=(levels, feats): x1 is the encoders' bf16 [8, M, 8] XCD-grouped tensor."""
    if in_dim is None:
        in_dim = weights[0].shape[1]
    out = _apply_decoder(_FusedMLP, x1, x2, x2_index, int(in_dim), out_act, mode, out_dtype, x1_grouped, *weights, *biases)
    if x2_packs is not None and x2 is not None and x2.requires_grad and out.grad_fn is not None:
        out.grad_fn.x2_packs = x2_packs      # (pack_start, ray_of_pack): x2_index is constant inside each pack (d/d x2 only)
    return out


# ------------------------------------------------------------------------------------------ ray march
def raymarch_ray(origins, dirs, dist_min, dist_max, num_samples, jitter=None, occupancy_bits=None, blas_level=7, want_ridx64=False):
    """'ray'-mode march + occupancy filter + pack.  Returns
    (ridx i32[M], pidx i32[M], samples f32[M,3], depths f32[M], deltas f32[M], boundary bool[M],
     pack_start i64[P+1], ray_of_pack i32[P])."""
    _check_gpu(origins, dirs)
    dev = origins.device
    N, S = origins.shape[0], int(num_samples)
    origins = origins.detach().contiguous().float()
    dirs = dirs.detach().contiguous().float()
    if jitter is None:
        jitter = torch.rand(N, S, device=dev)
    jitter = jitter.contiguous().float()
    tvals = _tvals(S, dev)
    counts = torch.empty(N, device=dev, dtype=torch.int32)
    occ = L.ptr(occupancy_bits) if occupancy_bits is not None else None
    st = L.stream()
    if N:
        _call("pag_raymarch_count", L.ptr(origins), L.ptr(dirs), N, S, L.ptr(tvals), L.ptr(jitter), float(dist_min),
                                       float(dist_max), occ, blas_level, L.ptr(counts), st)
    pack_start = torch.empty(N + 1, device=dev, dtype=torch.int64)      # [i] = first sample of ray i, [N] = M
    mailbox = _count_mailbox() if POLL_SAMPLE_COUNT else None
    if mailbox is not None:
        mailbox[1][0] = -1
    _call("pag_pack_offsets", L.ptr(counts), N, L.ptr(pack_start), mailbox[0].data_ptr() if mailbox is not None else None, st)
    # The pack kernel takes its write offsets from the device, so it is queued BEFORE the host learns the sample count:
    # buffers are sized for the N * S upper bound and trimmed to M afterwards.  The GPU then idles only for the read-back
    # itself instead of read-back + six allocations + four launches (0.11 ms per step).
    cap = N * S
    ridx = torch.empty(cap, device=dev, dtype=torch.int32)
    pidx = torch.empty(cap, device=dev, dtype=torch.int32)
    samples = torch.empty(cap, 3, device=dev)
    depths = torch.empty(cap, device=dev)
    deltas = torch.empty(cap, device=dev)
    boundary = torch.empty(cap, device=dev, dtype=torch.uint8)
    ridx64 = torch.empty(cap, device=dev, dtype=torch.int64) if want_ridx64 else None     # wisp hands out int64 ray ids
    if cap:
        _call("pag_raymarch_pack", L.ptr(origins), L.ptr(dirs), N, S, L.ptr(tvals), L.ptr(jitter), float(dist_min),
                                      float(dist_max), occ, blas_level, L.ptr(pack_start), L.ptr(ridx), L.ptr(pidx),
                                      L.ptr(samples), L.ptr(depths), L.ptr(deltas), L.ptr(boundary), L.ptr(ridx64), st)
    # one pack per RAY (empty packs allowed): no nonzero() / second host sync.  A ray without samples composites to the
    # background with alpha = depth = 0 and hit = False, exactly what the pre-filled buffers hold (Appendix E.10).
    M = _poll_count(mailbox) if mailbox is not None else -1
    if M < 0:
        M = int(pack_start[N].item())          # stream-synchronising read-back
        if mailbox is not None and int(mailbox[1][0]) < 0:
            # the kernel has finished (the read-back above waited for it) and its store never reached the mailbox: device
            # writes to this pinned allocation are not visible to the host on this system - stop polling for good
            _disable_polling()
    if mailbox is not None:
        if int(mailbox[1][0]) >= 0:       # the kernel has written its word: nobody else will touch it, back to the pool
            _release_mailbox(mailbox)
        # else: a store may still arrive later (timed-out poll) - the word is dropped, never lent again
    if want_ridx64:
        return (ridx[:M], pidx[:M], samples[:M], depths[:M], deltas[:M], boundary[:M].view(torch.bool), pack_start, _ray_iota(N, dev),
                ridx64[:M])
    return (ridx[:M], pidx[:M], samples[:M], depths[:M], deltas[:M], boundary[:M].view(torch.bool),   # kernel writes 0 / 1
            pack_start, _ray_iota(N, dev))


POLL_SAMPLE_COUNT = os.environ.get("PAG_NO_POLL") is None
_MAILBOX_LOCK = threading.Lock()
_MAILBOX_FREE = []          # pinned int64[1] words not lent to a march in flight


def _disable_polling():
    global POLL_SAMPLE_COUNT
    POLL_SAMPLE_COUNT = False


def _count_mailbox(dev=None):
    """(pinned i64[1] tensor, its numpy view) lent to ONE raymarch_ray call: the pack-offset kernel stores the sample count there
    (system-scope release) and the host polls it - it continues as soon as that tiny kernel has run instead of after a
    stream-synchronising copy behind the pack kernel (~50 us of the GPU-idle window at the head of every step).  Words come from a
    small pool under a lock and go back with _release_mailbox(), so two marches in flight (another stream or thread, e.g. an
    interactive render next to training) never preset and poll the same word."""
    with _MAILBOX_LOCK:
        if _MAILBOX_FREE:
            return _MAILBOX_FREE.pop()
    t = torch.empty(1, dtype=torch.int64).pin_memory()
    return (t, t.numpy())


def _release_mailbox(mailbox):
    with _MAILBOX_LOCK:
        if len(_MAILBOX_FREE) < 16:
            _MAILBOX_FREE.append(mailbox)


def _poll_count(mailbox, timeout_s=0.5):
    """Wait for the mailbox; -1 if nothing arrived in time (the caller then falls back to the synchronous read-back).  The first
    ~50 us are a tight spin (the usual case: the count arrives within 10-20 us); after that the loop yields the GIL with short
    sleeps so a delayed kernel (earlier work queued on the stream) does not starve other Python threads."""
    arr = mailbox[1]
    t0 = time.perf_counter()
    spins = 0
    while True:
        v = int(arr[0])
        if v >= 0:
            return v
        spins += 1
        if spins % 256 == 0:
            el = time.perf_counter() - t0
            if el > timeout_s:
                return -1
            if el > 50e-6:
                time.sleep(20e-6)


_IOTA = {}


def _one_pack_per_ray(ray_of_pack, N):
    """True when the pack table is the cached identity of raymarch_ray (pack i = ray i, empty packs included): the compositing
    kernels then write every ray's outputs themselves (an empty pack yields the background / zeros) and need no pre-filled buffers."""
    return ray_of_pack.shape[0] == N and _IOTA.get((N, str(ray_of_pack.device))) is ray_of_pack


def _ray_iota(N, dev):
    """arange(N) i32, cached per (N, device): ray_of_pack of the one-pack-per-ray layout (read-only by contract)."""
    key = (N, str(dev))
    if key not in _IOTA:
        if len(_IOTA) > 16:
            _IOTA.clear()
        _IOTA[key] = torch.arange(N, device=dev, dtype=torch.int32)
    return _IOTA[key]


def view_embed(dirs, n_freq, width):
    """f32 [R, width] = wisp PositionalEmbedder(-dirs) zero padded (pag_view_embed); no gradient (callers keep the tensor-op
    form when the directions are learnable)."""
    _check_gpu(dirs)
    d = dirs.detach().contiguous().float()
    out = torch.empty(d.shape[0], width, device=d.device)
    _call("pag_view_embed", L.ptr(d), d.shape[0], int(n_freq), int(width), L.ptr(out), L.stream())
    return out


class _ViewEmbed(torch.autograd.Function):
    """view_embed() with d / d dirs (pose optimisation: the view direction depends on the camera rotation, ba_pipeline.py:89-90):
    pag_view_embed forward - the SAME values as the gradient-free path - and pag_view_embed_bwd instead of the ~25 launches of the
    tensor-op form's forward + backward."""

    @staticmethod
    def forward(ctx, dirs, n_freq, width):
        d = dirs.detach().contiguous().float()
        ctx.save_for_backward(d)
        ctx.cfg = (int(n_freq), int(width))
        out = torch.empty(d.shape[0], width, device=d.device)
        _call("pag_view_embed", L.ptr(d), d.shape[0], int(n_freq), int(width), L.ptr(out), L.stream())
        return out

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        n_freq, width = ctx.cfg
        g = g.contiguous().float()
        out = torch.empty_like(d)
        _call("pag_view_embed_bwd", L.ptr(d), d.shape[0], n_freq, width, L.ptr(g), L.ptr(out), L.stream())
        return out, None, None


def view_embed_grad(dirs, n_freq, width):
    """view_embed() differentiable with respect to `dirs` (f32 [R,3] on the GPU)."""
    _check_gpu(dirs)
    return _apply(_ViewEmbed, dirs, n_freq, width)


class _PoseRays(torch.autograd.Function):
    """pc_nerf/ba_pipeline.py:85-92 as one launch each way (pag_pose_rays_fwd / _bwd): camera-frame rays -> world-frame rays through the
    camera parameters [C,9], differentiable with respect to the parameters."""

    @staticmethod
    def forward(ctx, params, cam, rays_per_entry, origins_c, dirs_c):
        p = params.detach().contiguous().float()
        oc, dc = origins_c.detach().contiguous().float(), dirs_c.detach().contiguous().float()
        N = oc.shape[0]
        ow, dw = torch.empty(N, 3, device=p.device), torch.empty(N, 3, device=p.device)
        _call("pag_pose_rays_fwd", L.ptr(p), p.shape[0], L.ptr(cam), int(rays_per_entry), L.ptr(oc), L.ptr(dc), N, L.ptr(ow), L.ptr(dw), L.stream())
        ctx.save_for_backward(p, cam, oc, dc)
        ctx.rpe = int(rays_per_entry)
        ctx.set_materialize_grads(False)
        return ow, dw

    @staticmethod
    def backward(ctx, g_o, g_d):
        p, cam, oc, dc = ctx.saved_tensors
        if g_o is None and g_d is None:
            return None, None, None, None, None
        g_o = g_o.contiguous().float() if g_o is not None else None
        g_d = g_d.contiguous().float() if g_d is not None else None
        d_params = torch.empty_like(p)
        ws_bytes = L.load().pag_pose_rays_bwd_workspace_bytes(p.shape[0])
        ws = torch.empty(ws_bytes // 4, device=p.device)
        _call("pag_pose_rays_bwd", L.ptr(p), p.shape[0], L.ptr(cam), ctx.rpe, L.ptr(oc), L.ptr(dc), oc.shape[0], L.ptr(g_o), L.ptr(g_d), L.ptr(d_params),
              L.ptr(ws), ws_bytes, L.stream())
        return d_params, None, None, None, None


def pose_rays(params, cam, rays_per_entry, origins_c, dirs_c):
    """-> (origins_w, dirs_w) f32 [N,3]: ray i through camera cam[i // rays_per_entry] (row of params f32 [C,9] = a1, a2, t); cam i32."""
    _check_gpu(params, cam, origins_c, dirs_c)
    if origins_c.shape != dirs_c.shape or origins_c.dim() != 2 or origins_c.shape[1] != 3:
        raise RuntimeError("pose_rays: origins / dirs must both be [N,3], got %s and %s" % (tuple(origins_c.shape), tuple(dirs_c.shape)))
    n_entries = (origins_c.shape[0] + rays_per_entry - 1) // rays_per_entry
    if cam.dtype != torch.int32 or cam.numel() < n_entries or params.dim() != 2 or params.shape[1] != 9:
        raise RuntimeError("pose_rays: cam must be int32 with one entry per %d rays, params [C,9]" % rays_per_entry)
    return _PoseRays.apply(params, cam.contiguous(), int(rays_per_entry), origins_c, dirs_c)


def pose_points(params, cam, rays_per_entry, origins_c, dirs_c, depth):
    """-> points f32 [N,3] (no gradient): sum_k (o_c - t + d_c * depth)[k] R[k] of ray i's camera cam[i // rays_per_entry] - utils/outlier_rejection.py:74-97."""
    _check_gpu(params, cam, origins_c, dirs_c, depth)
    N = origins_c.shape[0]
    if origins_c.shape != dirs_c.shape or origins_c.dim() != 2 or origins_c.shape[1] != 3 or depth.numel() != N:
        raise RuntimeError("pose_points: origins / dirs [N,3] and depth [N], got %s, %s, %s" % (tuple(origins_c.shape), tuple(dirs_c.shape), tuple(depth.shape)))
    n_entries = (N + rays_per_entry - 1) // rays_per_entry
    if cam.dtype != torch.int32 or cam.numel() < n_entries or params.dim() != 2 or params.shape[1] != 9:
        raise RuntimeError("pose_points: cam must be int32 with one entry per %d rays, params [C,9]" % rays_per_entry)
    prm, oc, dc = params.detach().contiguous().float(), origins_c.detach().contiguous().float(), dirs_c.detach().contiguous().float()
    dep = depth.detach().reshape(-1).contiguous().float()
    out = torch.empty(N, 3, device=oc.device)
    _call("pag_pose_points", prm.data_ptr(), prm.shape[0], cam.contiguous().data_ptr(), int(rays_per_entry), oc.data_ptr(), dc.data_ptr(), dep.data_ptr(), N,
          out.data_ptr(), L.stream())
    return out


class _RaySamples(torch.autograd.Function):
    """samples = origins[ray] + dirs[ray] * depth with the march kernel's values as the forward result (bit-identical to
    the non-differentiable path) and d/d origins, d/d dirs as per-ray segmented sums - what autograd gives through
    wisp's `torch.addcmul(origins[ridx], dirs[ridx], depth)` when the rays come from learnable extrinsics
    (ba_pipeline.py:85-92)."""

    @staticmethod
    def forward(ctx, origins, dirs, samples, depths, pack_start, ray_of_pack):
        ctx.save_for_backward(depths, pack_start, ray_of_pack)
        ctx.N = origins.shape[0]
        return samples.view_as(samples)

    @staticmethod
    def backward(ctx, g):
        depths, pack_start, ray_of_pack = ctx.saved_tensors
        N = ctx.N
        g = g.reshape(-1, 3).float().contiguous()
        P = ray_of_pack.shape[0]
        # [N,6]: d/d origin | d/d dir, one launch (pag_ray_sample_grad); every ray has a pack on the one-pack-per-ray layout
        seg = (torch.empty if (P and _one_pack_per_ray(ray_of_pack, N)) else torch.zeros)(N, 6, device=g.device)
        if P and g.shape[0]:
            _call("pag_ray_sample_grad", L.ptr(pack_start), L.ptr(ray_of_pack), P, L.ptr(g), L.ptr(depths.reshape(-1).float().contiguous()),
                  L.ptr(seg), L.stream())
        elif P:
            seg.zero_()
        return seg[:, :3], seg[:, 3:], None, None, None, None


def ray_samples(origins, dirs, samples, depths, pack_start, ray_of_pack, ridx=None):
    """Attach the pose gradient to march-kernel samples (samples [M,3] or [M',k,3], depths alike, one pack per ray).
    ridx (i32, the ray of every SAMPLE, as the march kernels write it): the result is tagged (`_pag_rays`) so that an encoder that interpolates exactly these
    samples can send its position gradient to origins / dirs directly, reduced per ray inside its gather pass (encode(..., rays=...))."""
    out = _RaySamples.apply(origins, dirs, samples, depths, pack_start, ray_of_pack)
    if ridx is not None and _one_pack_per_ray(ray_of_pack, origins.shape[0]) and ridx.dtype == torch.int32:
        out._pag_rays = (origins, dirs, depths, pack_start, ridx)
    return out


_TVALS = {}


def _tvals(S, dev):
    """linspace(0,1,S) computed on the CPU (bit-identical to the oracle's) and cached per (S, device)."""
    key = (S, str(dev))
    if key not in _TVALS:
        _TVALS[key] = torch.linspace(0, 1.0, S).to(dev)
    return _TVALS[key]


def occupancy_coarse(occupancy_bits, blas_level):
    """u32 bitfield of the (2^level / 4)^3 coarse occupancy the voxel march keeps in LDS, or None when the level has none."""
    nbytes = L.load().pag_occupancy_coarse_bytes(int(blas_level))
    if occupancy_bits is None or nbytes == 0:
        return None
    coarse = torch.empty(nbytes // 4, device=occupancy_bits.device, dtype=torch.int32)
    _call("pag_occupancy_coarse", L.ptr(occupancy_bits), int(blas_level), L.ptr(coarse), L.stream())
    return coarse


VOXEL_SCRATCH_MAX = 1 << 30      # bytes of nugget scratch raymarch_voxel() may allocate per call


def raymarch_voxel(origins, dirs, dist_min, dist_max, samples_per_voxel, occupancy_bits=None, blas_level=7, max_travel=None,
                   occupancy_coarse_bits=None, want_packs=False):
    """'voxel'-mode march (3-D DDA over the occupancy grid).  Returns per NUGGET ridx i32[M'], pidx i32[M'] and per sample
    samples f32[M',k,3], depths f32[M',k], deltas f32[M'*k], boundary bool[M'*k].
    max_travel: the tracer's travel filter (tracers/panoptic_packed_rf_tracer.py:88-108) applied inside the walk (None = off).
    want_packs: also (pack_start i64[N+1] in samples, ray_of_pack i32[N] = arange, ridx_sample i32[M'*k], ridx64 i64[M']) - one
    (possibly empty) pack per ray, straight from the kernels: no unique / nonzero / repeat_interleave passes, and the sample
    count reaches the host through the polled mailbox of raymarch_ray()."""
    return raymarch_voxel_finish(raymarch_voxel_begin(origins, dirs, dist_min, dist_max, samples_per_voxel, occupancy_bits, blas_level, max_travel,
                                                      occupancy_coarse_bits, want_packs))


def raymarch_voxel_begin(origins, dirs, dist_min, dist_max, samples_per_voxel, occupancy_bits=None, blas_level=7, max_travel=None,
                         occupancy_coarse_bits=None, want_packs=False):
    """First half of raymarch_voxel(): the walk (nugget counts / candidates) and the pack offsets are QUEUED, nothing is waited for.  -> state for
    raymarch_voxel_finish(), which reads the sample count, sizes the packed tensors and queues the expansion.  A caller with several marches to do
    (PanopticPackedRFTracer.render_packs) begins the next one before it finishes this one: the walk - a latency-bound DDA per ray - then runs while the
    host is busy elsewhere, and the count is there when it is asked for."""
    _check_gpu(origins, dirs)
    dev = origins.device
    N, k = origins.shape[0], int(samples_per_voxel)
    origins = origins.detach().contiguous().float()
    dirs = dirs.detach().contiguous().float()
    counts = torch.empty(N, device=dev, dtype=torch.int32)
    occ = L.ptr(occupancy_bits) if occupancy_bits is not None else None
    coarse = L.ptr(occupancy_coarse_bits) if (occupancy_coarse_bits is not None and occupancy_bits is not None) else None
    travel = float("inf") if max_travel is None else float(max_travel)
    st = L.stream()
    # one walk: pass 1 records the nuggets ([cap][N] scratch, 12 bytes each), pass 2 expands them in parallel; above VOXEL_SCRATCH_MAX
    # bytes of scratch the rays are walked twice instead (pag_raymarch_voxel_count / _pack)
    cap = int(L.load().pag_raymarch_voxel_nugget_capacity(blas_level))
    nug_t = nug_cell = None
    if N and 2 * cap * N * 12 <= VOXEL_SCRATCH_MAX:
        nug_t = torch.empty(2, cap, N, 2, device=dev)              # [0]: the walk's candidates [step][ray], [1]: the kept nuggets [ray][slot]
        nug_cell = torch.empty(2, cap, N, device=dev, dtype=torch.int32)
        _call("pag_raymarch_voxel_count_nuggets", L.ptr(origins), L.ptr(dirs), N, k, float(dist_min), float(dist_max), occ, coarse,
              blas_level, travel, L.ptr(counts), L.ptr(nug_t), L.ptr(nug_cell), st)
    elif N:
        _call("pag_raymarch_voxel_count", L.ptr(origins), L.ptr(dirs), N, k, float(dist_min), float(dist_max), occ, coarse, blas_level,
              travel, L.ptr(counts), st)
    pack_start = torch.empty(N + 1, device=dev, dtype=torch.int64)      # [i] = first SAMPLE of ray i, [N] = M' * k
    mailbox = _count_mailbox() if POLL_SAMPLE_COUNT else None
    if mailbox is not None:
        mailbox[1][0] = -1
    _call("pag_pack_offsets", L.ptr(counts), N, L.ptr(pack_start), mailbox[0].data_ptr() if mailbox is not None else None, st)
    return (origins, dirs, N, k, float(dist_min), float(dist_max), occupancy_bits, occupancy_coarse_bits, occ, coarse, blas_level, travel, counts, nug_t, nug_cell,
            pack_start, mailbox, want_packs)


def raymarch_voxel_finish(state):
    (origins, dirs, N, k, dist_min, dist_max, _occ_t, _coarse_t, occ, coarse, blas_level, travel, counts, nug_t, nug_cell, pack_start, mailbox, want_packs) = state
    dev = origins.device
    st = L.stream()
    total = _poll_count(mailbox) if mailbox is not None else -1
    if total < 0:
        total = int(pack_start[N].item())          # stream-synchronising read-back
        if mailbox is not None and int(mailbox[1][0]) < 0:
            _disable_polling()
    if mailbox is not None and int(mailbox[1][0]) >= 0:
        _release_mailbox(mailbox)
    Mn = total // k
    ridx = torch.empty(Mn, device=dev, dtype=torch.int32)
    pidx = torch.empty(Mn, device=dev, dtype=torch.int32)
    samples = torch.empty(Mn, k, 3, device=dev)
    depths = torch.empty(Mn, k, device=dev)
    deltas = torch.empty(Mn * k, device=dev)
    boundary = torch.empty(Mn * k, device=dev, dtype=torch.uint8)
    ridx_sample = torch.empty(Mn * k, device=dev, dtype=torch.int32) if want_packs else None
    ridx64 = torch.empty(Mn, device=dev, dtype=torch.int64) if want_packs else None
    if Mn and nug_t is not None:
        _call("pag_raymarch_voxel_pack_nuggets", L.ptr(origins), L.ptr(dirs), N, k, L.ptr(pack_start), L.ptr(nug_t), L.ptr(nug_cell), blas_level,
              L.ptr(ridx), L.ptr(pidx), L.ptr(samples), L.ptr(depths), L.ptr(deltas), L.ptr(boundary), L.ptr(ridx_sample), L.ptr(ridx64), st)
    elif Mn:
        _call("pag_raymarch_voxel_pack", L.ptr(origins), L.ptr(dirs), N, k, dist_min, dist_max, occ, coarse, blas_level,
              travel, L.ptr(pack_start), L.ptr(ridx), L.ptr(pidx), L.ptr(samples), L.ptr(depths), L.ptr(deltas), L.ptr(boundary),
              L.ptr(ridx_sample), L.ptr(ridx64), st)
    if want_packs:
        return (ridx, pidx, samples, depths, deltas, boundary.view(torch.bool), pack_start, _ray_iota(N, dev), ridx_sample, ridx64)
    return ridx, pidx, samples, depths, deltas, boundary.view(torch.bool)


class MarchBuffers:
    """Upper-bound packed-sample buffers of one march configuration, for the graph path (pagnerf_amd/graphs.py): the march kernels write
    into them without the host knowing the sample count, pad_to() appends inert samples up to a fixed capacity, and the first
    `capacity` elements are the STATIC tensors a captured HIP graph reads.  mode 'ray': per_ray = samples per ray (k = 1);
    mode 'voxel': per_ray = pag_raymarch_voxel_nugget_capacity(level), k = samples per nugget."""

    def __init__(self, mode, N, per_ray, k, dev):
        self.mode, self.N, self.k = mode, int(N), int(k)
        ne = self.N * int(per_ray)                 # entries (nuggets; = samples in 'ray' mode)
        self.cap = ne * self.k                     # samples
        self.samples = torch.zeros(self.cap, 3, device=dev)
        self.depths = torch.zeros(self.cap, device=dev)
        self.deltas = torch.zeros(self.cap, device=dev)
        self.boundary = torch.zeros(self.cap, device=dev, dtype=torch.uint8)
        self.ridx_entry = torch.zeros(ne, device=dev, dtype=torch.int32)
        self.ridx_sample = self.ridx_entry if self.k == 1 else torch.zeros(self.cap, device=dev, dtype=torch.int32)
        self.ridx64 = torch.zeros(ne, device=dev, dtype=torch.int64)
        self.pidx = torch.zeros(ne, device=dev, dtype=torch.int32)
        self.counts = torch.zeros(self.N, device=dev, dtype=torch.int32)
        self.pack_start = torch.zeros(self.N + 1, device=dev, dtype=torch.int64)
        # min(pack_start, capacity), written by pad_to(): the pack table of every launch that is queued on the capacity-sized views before
        # the host knows the sample count - no per-ray kernel can walk past the capacity when a batch overflows it
        self.pack_start_c = torch.zeros(self.N + 1, device=dev, dtype=torch.int64)
        if mode == "voxel":
            self.nug_t = torch.empty(2, int(per_ray), self.N, 2, device=dev)
            self.nug_cell = torch.empty(2, int(per_ray), self.N, device=dev, dtype=torch.int32)

    def pad_to(self, capacity):
        """Queue pag_pad_packed: samples [M, capacity) become filler samples outside every pack (pack_start is left alone) and
        pack_start_c = min(pack_start, capacity)."""
        assert 0 < capacity <= self.cap and capacity % self.k == 0
        _call("pag_pad_packed", L.ptr(self.pack_start), self.N, int(capacity), self.k, L.ptr(self.samples), L.ptr(self.depths), L.ptr(self.deltas),
              L.ptr(self.ridx_sample) if self.k > 1 else None, L.ptr(self.ridx_entry), L.ptr(self.ridx64), L.ptr(self.pidx), L.ptr(self.boundary),
              L.ptr(self.pack_start_c), L.stream())


def _offsets(buf, N, mb_ptr, st, pad_capacity, dirs, dirs_out):
    """The scan of the per-ray counts into the pack table, alone or fused with the padding / direction copy of the graph path."""
    if pad_capacity is None:
        _call("pag_pack_offsets", L.ptr(buf.counts), N, L.ptr(buf.pack_start), mb_ptr, st)
        return
    assert 0 < pad_capacity <= buf.cap and pad_capacity % buf.k == 0
    copy = dirs_out is not None and dirs_out.data_ptr() != dirs.data_ptr()
    _call("pag_pack_offsets_pad", L.ptr(buf.counts), N, L.ptr(buf.pack_start), mb_ptr, int(pad_capacity), buf.k, L.ptr(buf.samples), L.ptr(buf.depths),
          L.ptr(buf.deltas), L.ptr(buf.ridx_sample) if buf.k > 1 else None, L.ptr(buf.ridx_entry), L.ptr(buf.ridx64), L.ptr(buf.pidx),
          L.ptr(buf.boundary), L.ptr(buf.pack_start_c), L.ptr(dirs) if copy else None, L.ptr(dirs_out) if copy else None, st)


def march_into(buf, origins, dirs, dist_min, dist_max, num_samples, jitter=None, occupancy_bits=None, blas_level=7, max_travel=None,
               occupancy_coarse_bits=None, pad_capacity=None, dirs_out=None):
    """The ray march of raymarch_ray() / raymarch_voxel() written into `buf` (MarchBuffers) WITHOUT waiting for the sample count:
    count -> offsets (+ pinned mailbox) -> pack are queued back to back.  -> the mailbox (poll it with _poll_count() when convenient;
    give it back with _release_mailbox()) or None when polling is disabled (the caller then reads buf.pack_start[N] itself).
    pad_capacity: the offsets launch also pads the batch to that capacity (MarchBuffers.pad_to) and, with dirs_out, copies the ray
    directions into that static tensor - one launch instead of three (pag_pack_offsets_pad)."""
    _check_gpu(origins, dirs)
    dev = origins.device
    N = origins.shape[0]
    assert N == buf.N
    origins = origins.detach().contiguous().float()
    dirs = dirs.detach().contiguous().float()
    occ = L.ptr(occupancy_bits) if occupancy_bits is not None else None
    st = L.stream()
    mailbox = _count_mailbox() if POLL_SAMPLE_COUNT else None
    if mailbox is not None:
        mailbox[1][0] = -1
    mb_ptr = mailbox[0].data_ptr() if mailbox is not None else None
    if buf.mode == "ray":
        S = int(num_samples)
        if jitter is None:
            jitter = torch.rand(N, S, device=dev)
        jitter = jitter.contiguous().float()
        tvals = _tvals(S, dev)
        _call("pag_raymarch_count", L.ptr(origins), L.ptr(dirs), N, S, L.ptr(tvals), L.ptr(jitter), float(dist_min), float(dist_max), occ,
              blas_level, L.ptr(buf.counts), st)
        _offsets(buf, N, mb_ptr, st, pad_capacity, dirs, dirs_out)
        _call("pag_raymarch_pack", L.ptr(origins), L.ptr(dirs), N, S, L.ptr(tvals), L.ptr(jitter), float(dist_min), float(dist_max), occ,
              blas_level, L.ptr(buf.pack_start), L.ptr(buf.ridx_entry), L.ptr(buf.pidx), L.ptr(buf.samples), L.ptr(buf.depths),
              L.ptr(buf.deltas), L.ptr(buf.boundary), L.ptr(buf.ridx64), st)
    else:
        k = buf.k
        coarse = L.ptr(occupancy_coarse_bits) if (occupancy_coarse_bits is not None and occupancy_bits is not None) else None
        travel = float("inf") if max_travel is None else float(max_travel)
        _call("pag_raymarch_voxel_count_nuggets", L.ptr(origins), L.ptr(dirs), N, k, float(dist_min), float(dist_max), occ, coarse,
              blas_level, travel, L.ptr(buf.counts), L.ptr(buf.nug_t), L.ptr(buf.nug_cell), st)
        _offsets(buf, N, mb_ptr, st, pad_capacity, dirs, dirs_out)
        _call("pag_raymarch_voxel_pack_nuggets", L.ptr(origins), L.ptr(dirs), N, k, L.ptr(buf.pack_start), L.ptr(buf.nug_t), L.ptr(buf.nug_cell), blas_level,
              L.ptr(buf.ridx_entry), L.ptr(buf.pidx), L.ptr(buf.samples), L.ptr(buf.depths), L.ptr(buf.deltas), L.ptr(buf.boundary),
              L.ptr(buf.ridx_sample), L.ptr(buf.ridx64), st)
    return mailbox, jitter


def copy_batch(dsts, srcs):
    """dst[i].copy_(src[i]) for lists of same-shape, same-dtype contiguous GPU tensors as ONE launch per 16 tensors (pag_copy_batch);
    pairs that need a conversion or are not contiguous take torch's copy_."""
    easy, hard = [], []
    for d, s_ in zip(dsts, srcs):
        (easy if (d.dtype == s_.dtype and d.shape == s_.shape and d.is_contiguous() and s_.is_contiguous() and d.is_cuda and s_.is_cuda)
         else hard).append((d, s_))
    for c0 in range(0, len(easy), 16):
        part = easy[c0:c0 + 16]
        n = len(part)
        _call("pag_copy_batch", n, (ctypes.c_void_p * n)(*[d.data_ptr() for d, _ in part]), (ctypes.c_void_p * n)(*[s_.data_ptr() for _, s_ in part]),
              (ctypes.c_int64 * n)(*[d.numel() * d.element_size() for d, _ in part]), L.stream())
    for d, s_ in hard:
        d.copy_(s_)


def packs_from_boundary(ridx, boundary):
    """(pack_start i64[P+1], ray_of_pack i32[P]) from kaolin-style (ridx, boundary) arrays."""
    starts = torch.nonzero(boundary).reshape(-1)
    end = torch.tensor([boundary.shape[0]], device=boundary.device, dtype=torch.int64)
    return torch.cat([starts, end]).contiguous(), ridx[starts].int().contiguous()


def occupancy_update(density, occupancy, bits, decay, min_density):
    """In place: occupancy <- max(density, occupancy*decay); bits <- occupancy > min_density  (prune, nef :74-104).
    density f32 [cells] (any stride over dim 0), occupancy f32 [cells], bits i32 [ceil(cells/32)]."""
    _check_gpu(density, occupancy, bits)
    cells = occupancy.shape[0]
    density = density.detach()
    if density.dim() != 1 or density.dtype != torch.float32:
        density = density.reshape(cells, -1)[:, 0].float()
    assert occupancy.is_contiguous() and bits.is_contiguous() and bits.numel() * 32 >= cells
    _call("pag_occupancy_update", density.data_ptr(), density.stride(0), L.ptr(occupancy), L.ptr(bits), cells, float(decay),
          float(min_density), L.stream())


# ----------------------------------------------------------------------------------------- composite
class _Composite(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sigma, rgb, deltas, depths, pack_start, ray_of_pack, N, bg_white):
        _check_gpu(sigma, deltas)
        dev = sigma.device
        M = sigma.shape[0]
        P = ray_of_pack.shape[0]
        sigma = sigma.detach().contiguous().float()
        deltas = deltas.detach().contiguous().float()
        rgbc = rgb.detach().contiguous().float() if rgb is not None else None
        depc = depths.detach().contiguous().float() if depths is not None else None
        # padded batches (graphs.py): the weights of the filler samples past pack_start[P] are zeroed by the launch itself (n_samples)
        w = torch.empty(M, device=dev) if (P and M) else torch.zeros(M, device=dev)
        ctx.tail_zero = TAIL_ZERO
        if M and _one_pack_per_ray(ray_of_pack, N):      # the kernel writes every ray (background for empty packs): no fills
            alpha = torch.empty(N, device=dev)
            hit = torch.empty(N, device=dev, dtype=torch.uint8)
            out_rgb = torch.empty(N, 3, device=dev) if rgb is not None else None
            out_depth = torch.empty(N, device=dev) if depths is not None else None
        else:
            alpha = torch.zeros(N, device=dev)
            hit = torch.zeros(N, device=dev, dtype=torch.uint8)
            out_rgb = (torch.ones if bg_white else torch.zeros)(N, 3, device=dev) if rgb is not None else None
            out_depth = torch.zeros(N, device=dev) if depths is not None else None
        if P and M:                     # M == 0: every pack is empty, the outputs already hold the background
            _call("pag_composite_fwd", L.ptr(pack_start), L.ptr(ray_of_pack), P, L.ptr(sigma), L.ptr(deltas), L.ptr(depc),
                                          L.ptr(rgbc), L.BG_WHITE if bg_white else L.BG_BLACK, L.ptr(w), L.ptr(alpha),
                                          L.ptr(out_rgb), L.ptr(out_depth), L.ptr(hit), M if TAIL_ZERO else 0, L.stream())
        ctx.save_for_backward(sigma, rgbc, deltas, depc, pack_start, ray_of_pack, w, alpha)
        ctx.bg_white = bg_white
        ctx.mark_non_differentiable(hit, w)
        ctx.set_materialize_grads(False)      # unused outputs (alpha, depth, hit, w) arrive as None, not as zero-filled tensors
        return alpha, hit, out_rgb, out_depth, w

    @staticmethod
    def backward(ctx, g_alpha, _g_hit, g_rgb, g_depth, _g_w):
        sigma, rgbc, deltas, depc, pack_start, ray_of_pack, w, alpha = ctx.saved_tensors
        M, P = sigma.shape[0], ray_of_pack.shape[0]
        mk = torch.empty if (P and M) else torch.zeros          # samples covered by packs: the kernel writes every element; fillers: zeroed by the launch (n_samples)
        d_sigma = mk(M, device=sigma.device)
        d_rgb = mk(M, 3, device=sigma.device) if rgbc is not None else None
        gc = lambda t: t.contiguous().float() if t is not None else None
        g_alpha, g_rgb, g_depth = gc(g_alpha), gc(g_rgb), gc(g_depth)
        if P and M:
            _call("pag_composite_bwd", L.ptr(pack_start), L.ptr(ray_of_pack), P, L.ptr(sigma), L.ptr(deltas), L.ptr(depc),
                                          L.ptr(rgbc), L.BG_WHITE if ctx.bg_white else L.BG_BLACK, L.ptr(w), L.ptr(alpha),
                                          L.ptr(g_rgb), L.ptr(g_depth), L.ptr(g_alpha), L.ptr(d_sigma), L.ptr(d_rgb),
                                          M if ctx.tail_zero else 0, L.stream())
        return d_sigma, d_rgb, None, None, None, None, None, None


def composite(sigma, rgb, deltas, depths, pack_start, ray_of_pack, N, bg_white=True):
    """-> (alpha [N], hit u8 [N], rgb [N,3] | None, depth [N] | None, weights [M]); tracer :134-176."""
    return _apply(_Composite, sigma, rgb, deltas, depths, pack_start, ray_of_pack, N, bg_white)


class _CompositeFeats(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, weights, alpha, pack_start, ray_of_pack, N):
        _check_gpu(feats, weights, alpha)
        feats = feats.detach().contiguous()
        if feats.dtype not in (torch.float32, torch.bfloat16):
            feats = feats.float()
        C = feats.shape[1]
        P = ray_of_pack.shape[0]
        out = torch.zeros(N, C, device=feats.device)
        weights = weights.detach().contiguous()
        alpha = alpha.detach().contiguous()
        if P and feats.shape[0]:
            _call("pag_composite_feats_fwd", L.ptr(pack_start), L.ptr(ray_of_pack), P, L.ptr(weights), L.ptr(alpha),
                                                L.ptr(feats), L.dtype_code(feats), C, L.ptr(out), L.stream())
        ctx.save_for_backward(weights, alpha, pack_start, ray_of_pack)
        ctx.shape, ctx.fdtype = feats.shape, feats.dtype
        ctx.tail_zero = TAIL_ZERO
        return out

    @staticmethod
    def backward(ctx, g):
        weights, alpha, pack_start, ray_of_pack = ctx.saved_tensors
        M, C = ctx.shape
        P = ray_of_pack.shape[0]
        # every packed sample belongs to a pack, so the kernel writes every row: no zero fill of the [M,C] buffer
        d = torch.empty(M, C, device=weights.device, dtype=ctx.fdtype) if (P and not ctx.tail_zero) else torch.zeros(M, C, device=weights.device, dtype=ctx.fdtype)
        if P and M:
            _call("pag_composite_feats_bwd", L.ptr(pack_start), L.ptr(ray_of_pack), P, L.ptr(weights), L.ptr(alpha),
                  L.ptr(g.contiguous().float()), C, L.ptr(d), L.dtype_code(d), L.stream())
        return d, None, None, None, None, None


def composite_feats(feats, weights, alpha, pack_start, ray_of_pack, N):
    """out[ray] = alpha[ray] * sum_i w_i feats[i]  (weights/alpha detached; tracer :148-155,:197-205)."""
    return _apply(_CompositeFeats, feats, weights, alpha, pack_start, ray_of_pack, N)


class _CompositeFeatsWeights(torch.autograd.Function):
    """out[ray] = alpha_r * sum_i w_i f_i with w, alpha computed from (sigma, deltas) INSIDE the node and differentiable:
    the delta-density tracer composites the panoptic channels with the panoptic density's own weights and keeps their
    gradient (tracers/panoptic_dd_packed_rf_tracer.py:124-135,162-166).
    Backward: d f_i = alpha w_i G_r (composite_feats_bwd).  For sigma, sum_c G_rc out_rc = alpha_r sum_i w_i s_i with the
    per-sample scalar s_i = <G_r, f_i>, which is the black-background colour formula of pag_composite_bwd on the
    one-channel "colour" s - so d sigma comes from that kernel with rgb = (s,0,0) and upstream gradient (1,0,0)."""

    @staticmethod
    def forward(ctx, sigma, deltas, feats, ridx, pack_start, ray_of_pack, N):
        _check_gpu(sigma, deltas, feats)
        dev = sigma.device
        M, P = sigma.shape[0], ray_of_pack.shape[0]
        sigma = sigma.detach().contiguous().float()
        deltas = deltas.detach().contiguous().float()
        feats = feats.detach().contiguous()
        if feats.dtype not in (torch.float32, torch.bfloat16):
            feats = feats.float()
        C = feats.shape[1]
        w = torch.empty(M, device=dev)
        alpha = torch.zeros(N, device=dev)
        hit = torch.zeros(N, device=dev, dtype=torch.uint8)
        out = torch.zeros(N, C, device=dev)
        if P and M:
            _call("pag_composite_fwd", L.ptr(pack_start), L.ptr(ray_of_pack), P, L.ptr(sigma), L.ptr(deltas), None, None, L.BG_BLACK,
                  L.ptr(w), L.ptr(alpha), None, None, L.ptr(hit), 0, L.stream())
            _call("pag_composite_feats_fwd", L.ptr(pack_start), L.ptr(ray_of_pack), P, L.ptr(w), L.ptr(alpha), L.ptr(feats),
                  L.dtype_code(feats), C, L.ptr(out), L.stream())
        ctx.save_for_backward(sigma, deltas, feats, ridx, pack_start, ray_of_pack, w, alpha)
        ctx.mark_non_differentiable(alpha)
        return out, alpha

    @staticmethod
    def backward(ctx, g, _g_alpha):
        sigma, deltas, feats, ridx, pack_start, ray_of_pack, w, alpha = ctx.saved_tensors
        M, C = feats.shape
        P = ray_of_pack.shape[0]
        dev = sigma.device
        g = g.contiguous().float()
        d_feats = torch.zeros(M, C, device=dev, dtype=feats.dtype)
        d_sigma = torch.zeros(M, device=dev)
        if P and M:
            if ctx.needs_input_grad[2]:
                _call("pag_composite_feats_bwd", L.ptr(pack_start), L.ptr(ray_of_pack), P, L.ptr(w), L.ptr(alpha), L.ptr(g), C,
                      L.ptr(d_feats), L.dtype_code(d_feats), L.stream())
            if ctx.needs_input_grad[0]:
                s3 = torch.zeros(M, 3, device=dev)
                rl = ridx.long()
                step = max(1, (1 << 24) // max(C, 1))                        # bound the [chunk, C] temporary
                for lo in range(0, M, step):
                    hi = min(M, lo + step)
                    s3[lo:hi, 0] = (feats[lo:hi].float() * g[rl[lo:hi]]).sum(1)
                ones = torch.zeros(alpha.shape[0], 3, device=dev)
                ones[:, 0] = 1.0
                d_rgb = torch.empty(M, 3, device=dev)
                _call("pag_composite_bwd", L.ptr(pack_start), L.ptr(ray_of_pack), P, L.ptr(sigma), L.ptr(deltas), None, L.ptr(s3),
                      L.BG_BLACK, L.ptr(w), L.ptr(alpha), L.ptr(ones), None, None, L.ptr(d_sigma), L.ptr(d_rgb), 0, L.stream())
        return d_sigma, None, (d_feats if ctx.needs_input_grad[2] else None), None, None, None, None


def composite_features(sigma, deltas, feats, ridx, pack_start, ray_of_pack, N):
    """-> (out f32 [N,C], alpha f32 [N]) with gradients to sigma AND feats (delta-density tracer); ridx i32 [M]."""
    return _apply(_CompositeFeatsWeights, sigma, deltas, feats, ridx, pack_start, ray_of_pack, N)


HEAD_REBUILD = True      # wide softmax heads under head_composite(): statistics-only forward + rebuilt probabilities
CD_FUSED = os.environ.get("PAG_CD_FUSED", "1") != "0"      # density decoder + colour decoder in one launch (pag_mlp_fwd_args.x1_producer)
_NEXT_HOLD = None
_NEXT_PRODUCER = None


def decoder_hold(x1):
    """-> hold (dict): the next fused_mlp() call ON THE INPUT `x1` prepares its launch and parks it in `hold` instead of issuing it; hand `hold` to the
    decoder that consumes its output (colour_and_density(..., producer=hold)), which carries it in its own launch or issues it first, and call
    flush_hold(hold) afterwards in any case (issues the parked launch if nobody took it; forgets a hold no call picked up).  Single-threaded callers
    only, as everything behind the plugin API."""
    global _NEXT_HOLD
    _NEXT_HOLD = {"x1_ptr": x1.data_ptr()}
    return _NEXT_HOLD


def flush_hold(hold):
    global _NEXT_HOLD, _NEXT_PRODUCER
    if _NEXT_HOLD is hold:
        _NEXT_HOLD = None                # the call it was meant for never came
    if hold is not None and _NEXT_PRODUCER is hold:
        _NEXT_PRODUCER = None            # ... nor may a later, unrelated decoder pick this launch up as its producer
    if hold is not None and hold.get("args") is not None and not hold.get("taken"):
        _call("pag_mlp_fwd", ctypes.byref(hold["args"]), hold["M"], L.stream())
        hold["taken"] = True


HEAD_FWD_ONCE = os.environ.get("PAG_HEAD_FWD_ONCE", "1") != "0"      # wide softmax head: decoder + per-ray sum in one launch (0: statistics launch + pag_head_composite_fwd)
HEAD_FWD_ONCE_MIN_PER_RAY = int(os.environ.get("PAG_HEAD_FWD_ONCE_MIN_PER_RAY", "160"))      # average samples per ray from which the one-launch form is taken
TAIL_ZERO = False        # graphs.py: batches carry filler samples past pack_start[N] - per-sample tensors written pack by pack start as zeros
SAMPLES_HINT = None      # graphs.py: the REAL sample count expected in a padded batch (pag_head_composite_fwd picks its per-pack work split from it)


class _HeadComposite(_FusedMLP):
    """decoder (+ softmax) followed by the per-ray weighted sum of tracer :197-205, as ONE autograd node: the
    backward hands the decoder the gradient in rank-1 form (alpha * w_m * d out[ray]) so neither the [M,C] gradient
    nor a separate composite-backward launch exists."""

    @staticmethod
    def forward(ctx, x1, weights_w, alpha, ridx, pack_start, ray_of_pack, N, in_dim, out_act, out_dtype, grouped, *wb):
        # wide softmax head: where the library can, the decoder's launch forms the per-ray sums itself (pag_mlp_fwd_args.composite) - the output
        # tensor and the compositing arguments are prepared first and handed to _FusedMLP.forward through the ctx
        C = wb[len(wb) // 2 - 1].shape[0]
        M = x1.shape[1] if grouped is not None else x1.shape[0]
        P = ray_of_pack.shape[0]
        pre = None
        # (one wave per SIMD holds a ray's 112 partial sums next to the 112 logits: worth it where rays are long - the dense march, 16 tiles per ray:
        # 288 -> 260 us per 2.1 M samples; with the ~3 tiles per ray of the voxel march the two-launch form is faster, 375 against 434 us per 2.2 M)
        long_rays = (M if SAMPLES_HINT is None else SAMPLES_HINT) >= HEAD_FWD_ONCE_MIN_PER_RAY * P
        if HEAD_FWD_ONCE and long_rays and HEAD_REBUILD and C > 192 and P and M and out_act == L.ACT_SOFTMAX and out_dtype == torch.bfloat16:
            full = _one_pack_per_ray(ray_of_pack, N)
            pre = dict(pack_start=pack_start, ray_of_pack=ray_of_pack, weights=weights_w.detach().contiguous(), alpha=alpha.detach().contiguous(),
                       out=(torch.empty if full else torch.zeros)(N, C, device=x1.device))
            ctx.fwd_composite = pre
        probs = _HeadComposite._decode(ctx, x1, in_dim, out_act, out_dtype, grouped, *wb)
        ctx.fwd_composite = None
        if pre is not None and getattr(ctx, "composited", False):
            ctx.fwd_state = None
            ctx.hc = (pre["weights"], pre["alpha"], ridx)
            return pre["out"]
        return _HeadComposite._composite(ctx, probs, x1, weights_w, alpha, ridx, pack_start, ray_of_pack, N, grouped, *wb)

    @staticmethod
    def _decode(ctx, x1, in_dim, out_act, out_dtype, grouped, *wb):
        ctx.stats_only = HEAD_REBUILD
        ctx.rank1_expected = True            # the backward hands the decoder its gradient in rank-1 form (_backward_pair)
        return _FusedMLP.forward(ctx, x1, None, None, in_dim, out_act, L.MLP_MFMA_BF16, out_dtype, grouped, *wb)

    @staticmethod
    def _composite(ctx, probs, x1, weights_w, alpha, ridx, pack_start, ray_of_pack, N, grouped, *wb):
        C = wb[len(wb) // 2 - 1].shape[0]
        P = ray_of_pack.shape[0]
        M = x1.shape[1] if grouped is not None else x1.shape[0]
        # both compositing kernels used below write a row for every pack (zeros for an empty one)
        full = M and _one_pack_per_ray(ray_of_pack, N) and (probs is None or C <= 16)
        out = (torch.empty if full else torch.zeros)(N, C, device=x1.device)
        weights_w = weights_w.detach().contiguous()
        alpha = alpha.detach().contiguous()
        if probs is None:
            # wide softmax head: the forward wrote only the softmax statistics; the per-ray sums are formed from
            # probabilities rebuilt on the fly (pag_head_composite_fwd) - the [M, C] tensor never exists
            hidden_last, W_last, b_last, stats = ctx.fwd_state
            ctx.fwd_state = None
            if P and M:
                _call("pag_head_composite_fwd", L.ptr(pack_start), L.ptr(ray_of_pack), P, L.ptr(hidden_last), L.ptr(W_last),
                      L.ptr(b_last), C, L.ptr(stats), L.ptr(weights_w), L.ptr(alpha), L.ptr(out),
                      int(M if SAMPLES_HINT is None else SAMPLES_HINT), L.stream())
        elif P and M:
            _call("pag_composite_feats_fwd", L.ptr(pack_start), L.ptr(ray_of_pack), P, L.ptr(weights_w), L.ptr(alpha),
                  L.ptr(probs), L.dtype_code(probs), C, L.ptr(out), L.stream())
        ctx.hc = (weights_w, alpha, ridx)
        return out

    @staticmethod
    def backward(ctx, g):
        dx1, gwb = _HeadComposite._backward_pair(ctx, g, None)
        return (dx1, None, None, None, None, None, None, None, None, None, None, *gwb)

    @staticmethod
    def _backward_pair(ctx, g, dx1_into, wgrad_queue=None, **kw):
        weights_w, alpha, ridx = ctx.hc
        # upstream gradient in rank-1 form: alpha[ray] * w_m * g[ray] (detached weights, :148-155); the kernel forms the product
        grads = _FusedMLP._backward_impl(ctx, None, (g.contiguous().float(), weights_w.contiguous(), ridx.contiguous(), alpha.contiguous()),
                                         dx1_into, wgrad_queue=wgrad_queue, **kw)
        return grads[0], grads[8:]


class _SubCtx:
    """Per-head stand-in for the autograd ctx inside _HeadCompositePair."""

    def __init__(self, needs_x1, any_grad):
        self.needs_input_grad = (needs_x1, False)
        self.any_grad = any_grad          # does ANY input of the enclosing node (features, weights, biases) need a gradient?
        self.saved_tensors = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors


class _HeadCompositePair(torch.autograd.Function):
    """Two head_composite() decoders on the SAME grouped input (semantic + instance heads on the panoptic features) as one
    autograd node: the second decoder's backward adds its input gradient into the first one's tensor in place
    (pag_mlp_bwd_args.dx1_accumulate) instead of autograd summing two [8,M,8] tensors in a separate pass."""

    @staticmethod
    def forward(ctx, x1, weights_w, alpha, ridx, pack_start, ray_of_pack, N, in_dims, out_dtype, grouped, n_a, *wb):
        any_grad = any(ctx.needs_input_grad)
        sub_a, sub_b = _SubCtx(ctx.needs_input_grad[0], any_grad), _SubCtx(ctx.needs_input_grad[0], any_grad)
        part_a, part_b = wb[:n_a], wb[n_a:]
        # the second (narrow) decoder is prepared first and parked: where the library can, it is evaluated inside the first one's launch
        # (pag_mlp_fwd_args.pair: the features are read once); its compositing pass follows either way
        hold = {}
        sub_b.fwd_hold = hold
        probs_b = _HeadComposite._decode(sub_b, x1, in_dims[1], L.ACT_SOFTMAX, out_dtype, grouped, *part_b)
        sub_a.fwd_pair = hold
        out_a = _HeadComposite.forward(sub_a, x1, weights_w, alpha, ridx, pack_start, ray_of_pack, N, in_dims[0], L.ACT_SOFTMAX, out_dtype,
                                       grouped, *part_a)
        if hold.get("args") is not None and not hold.get("taken"):
            _call("pag_mlp_fwd", ctypes.byref(hold["args"]), hold["M"], L.stream())
        sub_a.fwd_pair = sub_b.fwd_hold = None
        out_b = _HeadComposite._composite(sub_b, probs_b, x1, weights_w, alpha, ridx, pack_start, ray_of_pack, N, grouped, *part_b)
        ctx.subs, ctx.n_a = [sub_a, sub_b], n_a
        return out_a, out_b

    @staticmethod
    def backward(ctx, g_a, g_b):
        sub_a, sub_b = ctx.subs
        # wide head first (it writes dx1), the narrow one accumulates.  Where the library can, the narrow head's backward rides in the wide
        # head's call (pag_mlp_bwd_args.pair: one read of the features, one write of the summed gradient): it is prepared first and parked.
        queue = []
        x1 = sub_a.saved_tensors[0]
        dx = torch.empty(x1.shape, device=x1.device, dtype=x1.dtype) if sub_a.needs_input_grad[0] else None
        hold = {}
        gb = _HeadComposite._backward_pair(sub_b, g_b, dx, queue, hold=hold if dx is not None else None)
        ga = _HeadComposite._backward_pair(sub_a, g_a, None, queue, dx1_out=dx, pair_hold=hold)
        if hold.get("args") is not None and not hold.get("taken"):          # not paired after all: the narrow head's own launch, after the wide one
            _call("pag_mlp_bwd", ctypes.byref(hold["args"]), hold["M"], L.stream())
        if queue:                                   # both heads' weight gradients: one narrow + one wide + one finish launch
            _launch_wgrad(queue, queue[0]["dz"].shape[0])
        return (ga[0], None, None, None, None, None, None, None, None, None, None, *ga[1], *gb[1])


def head_composite_pair(x1, heads, w, alpha, ridx, pack_start, ray_of_pack, N, out_dtype=torch.bfloat16, x1_grouped=None):
    """heads = ((weights, biases, in_dim), (weights, biases, in_dim)) -> (out_a, out_b), each as head_composite()."""
    (wa, ba, ia), (wb_, bb, ib) = heads
    return _apply_decoder(_HeadCompositePair, x1, w, alpha, ridx, pack_start, ray_of_pack, N, (int(ia), int(ib)), out_dtype, x1_grouped,
                                    len(wa) + len(ba), *wa, *ba, *wb_, *bb)


def head_composite(x1, weights, biases, w, alpha, ridx, pack_start, ray_of_pack, N, in_dim=None, out_act=L.ACT_NONE,
                   out_dtype=torch.bfloat16, x1_grouped=None):
    """alpha[ray] * sum_i w_i * act(decoder(x1))[i]  ->  f32 [N, out_dim]; ridx i32 [M] = ray of each sample."""
    if in_dim is None:
        in_dim = weights[0].shape[1]
    return _apply_decoder(_HeadComposite, x1, w, alpha, ridx, pack_start, ray_of_pack, N, int(in_dim), out_act, out_dtype, x1_grouped,
                                *weights, *biases)


# --------------------------------------------------------------------------------------- render loss
_LOSS_WS = {}


class NllTerm:
    """One `weight * mean(-log(prob[n, target_n] + eps) / temperature * conf_n)` term of render_loss."""

    def __init__(self, prob, target, weight=1.0, temperature=1.0, conf=None, mean_over="valid"):
        assert mean_over in ("valid", "all")
        self.prob, self.target, self.weight, self.temperature, self.conf, self.mean_over = prob, target, weight, temperature, conf, mean_over


class _RenderLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgb, prob_a, prob_b, rgb_gt, rgb_weight, term_a, term_b, eps):
        src = rgb if rgb is not None else (prob_a if prob_a is not None else prob_b)
        _check_gpu(src)
        dev = src.device
        N = src.shape[0]
        f = lambda t: t.detach().contiguous().float() if t is not None else None
        rgb_c, gt_c = f(rgb), f(rgb_gt)
        if rgb_c is not None:
            rgb_c, gt_c = rgb_c.reshape(N, 3), gt_c.reshape(N, 3)
        targs, keep = [], [rgb_c, gt_c]
        for prob, t in ((prob_a, term_a), (prob_b, term_b)):
            if prob is None:
                targs += [None, 0, None, None, 0.0, 1.0, 0]
                continue
            p = f(prob).reshape(N, -1)
            tg = t.target.detach().reshape(-1).long().contiguous()
            cf = f(t.conf).reshape(-1) if t.conf is not None else None
            keep += [p, tg, cf]
            targs += [L.ptr(p), p.shape[1], L.ptr(tg), L.ptr(cf), float(t.weight), 1.0 / float(t.temperature), int(t.mean_over == "all")]
        key = (dev.type, dev.index)
        if key not in _LOSS_WS:
            _LOSS_WS[key] = torch.zeros(L.load().pag_render_loss_workspace_bytes(), device=dev, dtype=torch.uint8)
        out = torch.empty(6, device=dev)
        head = [L.ptr(rgb_c), L.ptr(gt_c), N, float(rgb_weight)]
        _call("pag_render_loss_fwd", *head, *targs, float(eps), L.ptr(_LOSS_WS[key]), L.ptr(out), L.stream())
        ctx.call = (head, targs, float(eps))
        ctx.keep = keep                      # plain tensors (detached copies / inputs), not autograd-tracked
        ctx.out = out
        ctx.shapes = tuple(None if t is None else (t.shape, t.dtype) for t in (rgb, prob_a, prob_b))
        ctx.mark_non_differentiable(out)
        ctx.set_materialize_grads(False)
        return out[0], out

    @staticmethod
    def backward(ctx, g, _g_terms):
        head, targs, eps = ctx.call
        dev = ctx.out.device
        if g is None:
            return (None,) * 8
        grads = []
        for need, sh in zip(ctx.needs_input_grad[:3], ctx.shapes):
            grads.append(torch.empty(sh[0], device=dev) if (need and sh is not None) else None)
        g = g.detach().float().contiguous()
        _call("pag_render_loss_bwd", L.ptr(g), L.ptr(ctx.out), *head, *targs, eps, L.ptr(grads[0]), L.ptr(grads[1]), L.ptr(grads[2]), L.stream())
        grads = [None if d is None else (d if sh[1] == torch.float32 else d.to(sh[1])) for d, sh in zip(grads, ctx.shapes)]
        return grads[0], grads[1], grads[2], None, None, None, None, None


def render_loss(rgb=None, rgb_gt=None, rgb_weight=1.0, term_a=None, term_b=None, eps=1e-27):
    """-> (loss 0-dim, terms f32 [6] = total, rgb, A, B, denom A, denom B - detached).  The trainer's per-ray objective
    (pc_nerf/trainer.py:443-446, :459-465, loss/lin_assignment_things.py:80) as one launch forward and one backward."""
    if rgb is None and term_a is None and term_b is None:
        raise ValueError("render_loss needs at least one term")
    return _RenderLoss.apply(rgb, term_a.prob if term_a is not None else None, term_b.prob if term_b is not None else None,
                             rgb_gt, rgb_weight, term_a, term_b, eps)
