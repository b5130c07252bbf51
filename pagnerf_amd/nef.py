"""PanopticDeltaNeF on the gfx950 kernels - same constructor keywords, channels and output
shapes as pc_nerf/panoptic_delta_nef.py::PanopticDeltaNeF (+ its base pc_nerf/panoptic_nef.py),
so it can be registered under pc_nerf/trainer.py in place of the reference class.

Op order of rgb_semantics() follows pc_nerf/panoptic_delta_nef.py:155-259:
  grid.interpolate -> * lod_weights -> decoder_density -> relu(ch 0) -> decoder_color on
  cat(density_feats[16], PE(-ray_d)[27]) -> sigmoid ; delta_grid.interpolate(coords.detach())
  -> * lod_weights ; panoptic feats = feats.detach() + delta ; decoder_semantics / decoder_inst
  -> [sigmoid] -> [normalize] -> [/T] -> [softmax].
What is different is only WHERE things run: lod_weights is folded into the encode kernel, each
decoder (+ its sigmoid / softmax) is one fused launch, and the view embedding is computed once per
ray and gathered through ridx instead of being repeated per sample.
"""
import copy
import inspect
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as L
from . import ops
from .grids import HashGridHIP, PermutoGridHIP

_GRIDS = {"HashGridTorch": HashGridHIP, "HashGridTinyCudaNN": HashGridHIP, "HashGrid": HashGridHIP,
          "HashGridHIP": HashGridHIP, "PermutoGrid": PermutoGridHIP, "PermutoGridHIP": PermutoGridHIP}


class BasicDecoder(nn.Module):
    """wisp BasicDecoder's parameter layout (layers[i], lout) with a fused forward."""

    def __init__(self, input_dim, output_dim, num_layers=1, hidden_dim=64, bias=True):
        super().__init__()
        if hidden_dim != 64 or num_layers not in (1, 2):
            raise NotImplementedError("fused decoder supports hidden_dim=64 and 1-2 hidden layers (best.yaml); got %d x %d"
                                      % (num_layers, hidden_dim))
        self.input_dim, self.output_dim = input_dim, output_dim
        self.layers = nn.ModuleList([nn.Linear(input_dim if i == 0 else hidden_dim, hidden_dim, bias=bias)
                                     for i in range(num_layers)])
        self.lout = nn.Linear(hidden_dim, output_dim, bias=bias)

    def weights(self):
        lins = list(self.layers) + [self.lout]
        return [l.weight for l in lins], [l.bias for l in lins]

    def forward(self, x1, x2=None, x2_index=None, out_act=L.ACT_NONE, mode=L.MLP_MFMA_BF16, out_dtype=torch.float32,
                x1_grouped=None, x2_packs=None):
        W, b = self.weights()
        return ops.fused_mlp(x1, W, b, x2=x2, x2_index=x2_index, in_dim=self.input_dim, out_act=out_act, mode=mode,
                             out_dtype=out_dtype, x1_grouped=x1_grouped, x2_packs=x2_packs)


_PARAM_NAMES = {}


def _param_names(fn):
    """Parameter names of a channel function (inspect.signature costs ~40 us per call; the functions never change)."""
    key = getattr(fn, "__func__", fn)
    names = _PARAM_NAMES.get(key)
    if names is None:
        names = _PARAM_NAMES[key] = frozenset(inspect.signature(fn).parameters)
    return names


def positional_embed(x, num_freq):
    """wisp PositionalEmbedder: cat(x, sin(x*2^k), cos(x*2^k)), frequency-major."""
    bands = 2.0 ** torch.linspace(0.0, num_freq - 1, num_freq, device=x.device)
    w = (x[:, None, :] * bands[None, :, None]).reshape(x.shape[0], num_freq * x.shape[-1])
    return torch.cat([x, torch.sin(w), torch.cos(w)], dim=-1)


class PanopticDeltaNeF(nn.Module):
    accepts_ray_index = True     # the tracer may pass (ridx, ray_dirs) instead of a per-sample ray_d

    def __init__(self, grid_type="PermutoGrid", interpolation_type="linear", multiscale_type="cat", feature_dim=2,
                 num_lods=24, base_lod=2, hidden_dim=64, num_layers=1, activation_type="relu", layer_type="none",
                 embedder_type="positional", view_multires=4, pos_multires=4, position_input=False,
                 num_classes=-1, num_instances=-1,
                 sem_activation_type=None, sem_num_layers=None, sem_hidden_dim=None, sem_normalize=False,
                 sem_softmax=False, sem_sigmoid=False, sem_detach=True,
                 inst_num_layers=None, inst_hidden_dim=None, inst_normalize=False, inst_softmax=False,
                 inst_sigmoid=False, inst_detach=True, panoptic_features_type=None,
                 delta_num_layers=1, delta_hidden_dim=64, inst_soft_temperature=0.0,
                 raymarch_type="ray", precision="bf16", **kwargs):
        super().__init__()
        if activation_type != "relu" or (sem_activation_type or "relu") != "relu":
            raise NotImplementedError("fused decoders implement ReLU hidden activations (best.yaml)")
        if position_input:
            raise NotImplementedError                                   # panoptic_delta_nef.py:177
        if panoptic_features_type not in (None, "delta", "separate", "appearance", "pos_encoding", "position"):
            raise ValueError('Panoptic feature type "%s" not implemented for PanopticDeltaNeF' % panoptic_features_type)   # :234
        if multiscale_type not in ("cat", "sum"):
            raise NotImplementedError("'%s' not supported by this neural field. supported options ['cat', 'sum']" % multiscale_type)
        self.grid_type, self.multiscale_type = grid_type, multiscale_type
        self.feature_dim, self.num_lods, self.base_lod = feature_dim, num_lods, base_lod
        self.hidden_dim, self.num_layers = hidden_dim, num_layers
        self.view_multires, self.embedder_type = view_multires, embedder_type
        self.num_classes, self.num_instances = num_classes, num_instances
        self.sem_normalize, self.sem_softmax, self.sem_sigmoid = sem_normalize, sem_softmax, sem_sigmoid
        self.inst_normalize, self.inst_softmax, self.inst_sigmoid = inst_normalize, inst_softmax, inst_sigmoid
        self.panoptic_features_type = panoptic_features_type
        self.inst_soft_temperature = inst_soft_temperature
        self.raymarch_type = raymarch_type
        self.kwargs = kwargs
        self.set_precision(precision)
        # ---- grids (panoptic_nef.py:184-196, panoptic_delta_nef.py:39-44)
        gkw = dict(kwargs)
        gkw.pop("num_lods", None)
        if grid_type == "HashGridTinyCudaNN":
            gkw.setdefault("half_coords", True)       # grids/hash_grid_tinycudann.py:36 (custom_fwd cast_inputs=torch.half)
        self.grid = _GRIDS[grid_type](feature_dim, base_lod=base_lod, num_lods=num_lods,
                                      interpolation_type=interpolation_type, multiscale_type="cat", **gkw)
        self.lod_weights = torch.ones(num_lods * feature_dim)
        if panoptic_features_type in ("delta", "separate", None):
            self.delta_grid = copy.deepcopy(self.grid)
            if isinstance(self.delta_grid, PermutoGridHIP) and panoptic_features_type in ("delta", "separate") \
                    and "delta_capacity_log_2" in kwargs:
                self.delta_grid.set_capacity(kwargs["delta_capacity_log_2"])
        # ---- embedders (panoptic_nef.py:72-77; the position embedder of the panoptic branch: panoptic_delta_nef.py:46-53, always positional)
        self.view_embed_dim = 3 + 6 * view_multires if embedder_type == "positional" else 3
        self.pos_multires, self.pos_embed_dim = pos_multires, 3 + 6 * pos_multires
        # ---- decoders (panoptic_nef.py:85-164): the panoptic heads read the grid features, the embedded position or the raw position
        eff = feature_dim * num_lods if multiscale_type == "cat" else feature_dim
        pan_dim = {"position": 3, "pos_encoding": self.pos_embed_dim}.get(panoptic_features_type, eff)
        self.decoder_density = BasicDecoder(eff, 16, num_layers, hidden_dim)
        with torch.no_grad():
            self.decoder_density.lout.bias[0] = 1.0                      # panoptic_nef.py:123
        self.decoder_color = BasicDecoder(16 + self.view_embed_dim, 3, num_layers + 1, hidden_dim)
        self.decoder_semantics = BasicDecoder(pan_dim, num_classes, sem_num_layers or num_layers, sem_hidden_dim or hidden_dim)
        assert num_instances > 2
        self.decoder_inst = BasicDecoder(pan_dim, num_instances, inst_num_layers or num_layers, inst_hidden_dim or hidden_dim)
        self._fns = [(self.rgb_semantics, {"density", "rgb", "semantics", "inst_embedding"})]

    # -------------------------------------------------------------------------------- configuration
    def set_precision(self, precision):
        """'bf16': bf16 MFMA decoders on bf16 features (production); 'fp32': fp32 FMA-chain parity path."""
        assert precision in ("bf16", "fp32")
        self.precision = precision
        self.mlp_mode = L.MLP_MFMA_BF16 if precision == "bf16" else L.MLP_FP32
        self.feat_dtype = torch.bfloat16 if precision == "bf16" else torch.float32

    @property
    def device(self):
        return self.decoder_density.lout.weight.device

    def get_nef_type(self):
        return "delta_panoptic_nef"

    def get_supported_channels(self):
        s = set()
        for _, c in self._fns:
            s |= c
        return s

    # ----------------------------------------------------------------------------------- dispatcher
    def forward(self, channels=None, **kwargs):
        """wisp BaseNeuralField.forward semantics: str -> tensor, list -> list, set -> dict."""
        kwargs["compute_channels"] = channels                           # panoptic_nef.py:239-242
        req = {channels} if isinstance(channels, str) else set(channels)
        unsupported = req - self.get_supported_channels()
        if unsupported:
            raise Exception("Channels %s are not supported in %s" % (unsupported, type(self).__name__))
        out = {}
        for fn, chans in self._fns:
            if not (chans & req):
                continue
            params = _param_names(fn)
            res = fn(**{k: v for k, v in kwargs.items() if k in params})
            for c in chans & req:
                out[c] = res[c]
        if isinstance(channels, str):
            return out[channels]
        if isinstance(channels, list):
            return [out[c] for c in channels]
        return out

    # ------------------------------------------------------------------------------------ hot path
    def _view_embedding(self, ray_d, ridx, ray_dirs):
        """[R, 32] fp32 view embedding (27 used, zero padded) + int32 row index per sample."""
        if ray_dirs is not None and ridx is not None:
            src, index = ray_dirs, ridx
        else:
            src, index = ray_d, torch.arange(ray_d.shape[0], device=ray_d.device, dtype=torch.int32)
        if self.embedder_type == "positional" and src.is_cuda:
            width = 3 + 6 * self.view_multires
            width += (-width) % 8
            if src.requires_grad and torch.is_grad_enabled():                                # pose optimisation: d / d dirs through pag_view_embed_bwd
                return ops.view_embed_grad(src, self.view_multires, width), index
            return ops.view_embed(src, self.view_multires, width), index                     # one launch (pag_view_embed)
        pe = positional_embed(-src.float(), self.view_multires) if self.embedder_type == "positional" else -src.float()
        return F.pad(pe, (0, (-pe.shape[1]) % 8)).contiguous(), index

    def _grouped(self):
        """(levels, feats) when the bf16 path can use the XCD-grouped feature layout, else None."""
        if self.precision == "bf16" and self.multiscale_type == "cat" and ops.xcd8_supported(self.num_lods, self.feature_dim):
            return (self.num_lods, self.feature_dim)
        return None

    def _pan_grouped(self):
        """The panoptic heads' input layout: the grid features' (see _grouped) unless they read positions."""
        return None if self.panoptic_features_type in ("pos_encoding", "position") else self._grouped()

    @staticmethod
    def _pad8(x, dtype):
        """[M,k] -> [M, k rounded up to 8] in the decoders' input dtype (the fused decoders take k % 8 == 0 and ignore columns >= in_dim)."""
        return F.pad(x, (0, (-x.shape[1]) % 8)).to(dtype).contiguous()

    def _lod_weights_or_none(self):
        """None while every weight is 1 (the kernels then skip the multiply); the CPU-side check (three tensor ops, ~40 us -
        paid while the GPU waits for the step's first encode launch) is cached against the tensor's version counter."""
        lw = self.lod_weights
        key = (id(lw), lw._version)
        if getattr(self, "_lw_key", None) != key:
            self._lw_key, self._lw_ones = key, bool((lw == 1).all())
        return None if self._lw_ones else lw

    def _interp(self, grid, coords, addend=None):
        lw = self._lod_weights_or_none()
        feats = grid.interpolate_scaled(coords, lw, out_dtype=self.feat_dtype, layout="xcd8" if self._grouped() else None,
                                        addend=addend)
        if self.multiscale_type == "sum":                                          # :172-173, :221-222: levels summed, F columns left
            feats = self._pad8(feats.float().reshape(-1, self.num_lods, self.feature_dim).sum(-2), self.feat_dtype)
        return feats

    def prefetch_features(self, coords):
        """Queue the main grid's interpolation of `coords` NOW; rgb_semantics() called with the same tensor picks it up.
        The tracer calls this as soon as the ray march has returned: the GPU sits idle from the moment the host learns the
        sample count until the step's first encode launch, and everything the tracer and the nef dispatcher do in between
        (~35 us of Python) would otherwise be spent with an empty queue.  Same kernel, same arguments, same autograd node."""
        if self.multiscale_type != "cat" or self.grid is None:
            return
        self._prefetched = (coords, self._interp(self.grid, coords), torch.is_grad_enabled())

    def _panoptic_feats(self, feats_detached, coords):
        """:210-236 - `feats.detach() + delta` ('delta'), delta alone ('separate') or the main features ('appearance').
        On the grouped bf16 path the sum is formed inside the delta grid's encode launch (same rounding as the tensor add)."""
        t = self.panoptic_features_type
        if t == "pos_encoding":                                                    # :231 (the coordinates are NOT detached here)
            return self._pad8(positional_embed(coords.reshape(-1, 3).float(), self.pos_multires), self.feat_dtype)
        if t == "position":                                                        # :233
            return self._pad8(coords.reshape(-1, 3).float(), self.feat_dtype)
        if t == "appearance":
            return feats_detached
        if t == "separate":
            return self._interp(self.delta_grid, coords.detach())
        if self._grouped() is not None:
            return self._interp(self.delta_grid, coords.detach(), addend=feats_detached)
        return feats_detached + self._interp(self.delta_grid, coords.detach())

    def rgb_semantics(self, coords, ray_d=None, compute_channels=None, pidx=None, lod_idx=None, ridx=None, ray_dirs=None,
                      ray_packs=None):
        out = {}
        if not compute_channels:
            return out
        if isinstance(compute_channels, str):
            compute_channels = {compute_channels}
        batch, num_samples, _ = coords.shape
        mode = self.mlp_mode
        pre = getattr(self, "_prefetched", None)
        self._prefetched = None
        if pre is not None and pre[0] is coords and pre[2] == torch.is_grad_enabled():
            feats = pre[1]                                    # launched by the tracer right after the ray march
        else:
            feats = self._interp(self.grid, coords)                                   # :170-171
        # reused by panoptic_composited() for the same samples.  Detached unless a head reads the LIVE features (PanopticNeF with a detach flag
        # off): the cache must not pin the encode autograd node and its saved tensors beyond the trace - across the optimiser step, inside
        # a captured step's memory pool
        self._feat_cache = (coords, feats if self._heads_read_live_features() else feats.detach())
        grp = self._grouped()
        # with the colour decoder to follow, the density decoder's launch is parked and rides in the colour decoder's (ops.decoder_hold)
        hold = ops.decoder_hold(feats) if (ops.CD_FUSED and "rgb" in compute_channels and feats.is_cuda) else None
        # everything between decoder_hold() and flush_hold() sits in ONE try / finally: whatever raises in between (the density decoder, the view
        # embedding, the colour decoder before it took the parked launch), the launch is issued or forgotten and no later trace can inherit it
        try:
            density_feats = self.decoder_density(feats, mode=mode, out_dtype=self.feat_dtype, x1_grouped=grp)     # :184
            self._density_feats = density_feats.detach()             # the delta-density variant adds to its (detached) column 0 (pre-ReLU)
            if "rgb" in compute_channels:                                                 # :188, :196-204
                if num_samples != 1 and ridx is None:                                  # one direction per pack entry -> per sample
                    ray_d = ray_d[:, None].repeat(1, num_samples, 1).reshape(-1, 3)
                pe, index = self._view_embedding(ray_d, ridx, ray_dirs)
                W, b = self.decoder_color.weights()
                # colour decoder and the density column of its input as one node (ops._ColourDensity)
                rgb, sigma = ops.colour_and_density(density_feats, W, b, pe, index, self.decoder_color.input_dim, out_act=L.ACT_SIGMOID,
                                                    mode=mode, x2_packs=ray_packs if ridx is not None else None, producer=hold)
                density = sigma.reshape(batch, num_samples, 1)
                out["rgb"] = rgb.reshape(batch, num_samples, 3)
            else:
                density = torch.relu(density_feats[:, 0:1].float()).reshape(batch, num_samples, 1)   # :188
        finally:
            ops.flush_hold(hold)
        if "density" in compute_channels:
            out["density"] = density
        if "semantics" in compute_channels or "inst_embedding" in compute_channels:    # :210-236
            (sem_in, sem_grp), (inst_in, inst_grp) = self._head_inputs(feats, coords, compute_channels)
            if "semantics" in compute_channels:                                        # :238-244
                plain = not (self.sem_sigmoid or self.sem_normalize)
                act = L.ACT_SOFTMAX if (self.sem_softmax and plain) else L.ACT_NONE
                s = self.decoder_semantics(sem_in, out_act=act, mode=mode, x1_grouped=sem_grp, out_dtype=self.feat_dtype)
                if not plain:
                    s = torch.sigmoid(s) if self.sem_sigmoid else s
                    s = F.normalize(s, dim=-1) if self.sem_normalize else s
                    s = F.softmax(s, dim=-1) if self.sem_softmax else s
                out["semantics"] = s
            if "inst_embedding" in compute_channels:                                   # :246-257
                out["inst_embedding"] = self._inst_head(inst_in, inst_grp, mode)
        return out

    def _inst_head(self, inst_in, inst_grp, mode):
        """panoptic_delta_nef.py:246-257: decoder -> [sigmoid] -> [normalize] -> [/T] -> [softmax]."""
        plain = not (self.inst_sigmoid or self.inst_normalize or self.inst_soft_temperature > 0.0)
        act = L.ACT_SOFTMAX if (self.inst_softmax and plain) else L.ACT_NONE
        e = self.decoder_inst(inst_in, out_act=act, mode=mode, x1_grouped=inst_grp, out_dtype=self.feat_dtype)
        if not plain:
            e = torch.sigmoid(e) if self.inst_sigmoid else e
            e = F.normalize(e, dim=-1) if self.inst_normalize else e
            e = e / self.inst_soft_temperature if self.inst_soft_temperature > 0.0 else e
            e = F.softmax(e, dim=-1) if self.inst_softmax else e
        return e

    def _heads_read_live_features(self):
        """Does a panoptic head's gradient flow into the main grid?  Never here (panoptic_delta_nef.py:214,226: feats.detach())."""
        return False

    def _head_inputs(self, feats, coords, channels):
        """((semantic head's input, its grouped layout or None), (instance head's ...)) from the LIVE main features `feats`
        (panoptic_delta_nef.py:210-236: both heads read one tensor built from feats.detach())."""
        pan = self._panoptic_feats(feats.detach(), coords)
        grp = self._pan_grouped()
        return (pan, grp), (pan, grp)

    WIDE_HEAD_PAD = os.environ.get("PAG_WIDE_HEAD_PAD", "1") != "0"

    def _wide_head_weights(self, dec, x):
        """(weights, biases) of a wide softmax head for the fused kernels.  The dedicated wide-softmax kernels (one-launch forward with the per-ray sum,
        statistics-only forward, block-wise backward: include/pagnerf_hip.h) are written for best.yaml's THREE-layer 200-way head (`inst_num_layers: 2`,
        pc_nerf/panoptic_nef.py:157-164).  The two-layer head of `inst_num_layers: 1` (configs/bup20/config_hp_base.yaml:71, lin_assign_delta_app.yaml:120 and
        two more YAMLs) reaches them as the same network with an IDENTITY middle layer: h1 = relu(I h0 + 0) = h0 bit for bit (h0 is a ReLU output already
        rounded to bf16; every product is h0[k] x 1 or x 0 and the fp32 accumulator holds h0[j] exactly), and backwards dz0 = (I^T dz1) masked by h0 > 0 - the
        mask dz1 already carries - so the values are those of the two-layer network at ~12 % more decoder arithmetic, instead of the generic kernels'
        +23 % of the whole step.  The identity and the zero bias are constants (no gradient; what the kernels form for them is dropped).
        PAG_WIDE_HEAD_PAD=0 keeps the generic path."""
        W, b = dec.weights()
        if not (self.WIDE_HEAD_PAD and len(W) == 2 and x.is_cuda and self.precision == "bf16" and W[0].shape[0] == 64 and 192 < W[1].shape[0] <= 224
                and b[0] is not None and b[1] is not None):
            return W, b
        pad = getattr(self, "_wide_pad", None)
        if pad is None or pad[0].device != W[0].device:
            pad = self._wide_pad = (torch.eye(64, device=W[0].device), torch.zeros(64, device=W[0].device))
        return [W[0], pad[0], W[1]], [b[0], pad[1], b[1]]

    def can_fuse_panoptic(self, channels):
        """True when the semantic / instance heads can run as decoder + compositing in one autograd node."""
        if self.precision != "bf16" or self._pan_grouped() is None:
            return False
        ok = True
        if "semantics" in channels:
            ok &= self.sem_softmax and not (self.sem_sigmoid or self.sem_normalize)
        if "inst_embedding" in channels:
            ok &= self.inst_softmax and not (self.inst_sigmoid or self.inst_normalize or self.inst_soft_temperature > 0.0)
        return bool(ok)

    def panoptic_composited(self, coords, channels, w, alpha, ridx, pack_start, ray_of_pack, N):
        """Composited panoptic channels [N, C] for the packed samples `coords` ([M,1,3] or [M',k,3]): same arithmetic as
        rgb_semantics() :210-255 followed by tracer :197-205, but each head + its per-ray weighted sum is one autograd
        node whose backward feeds the decoder a rank-1 gradient (ops.head_composite)."""
        cache = getattr(self, "_feat_cache", None)
        self._feat_cache = None                               # one use: nothing of this trace stays referenced from the module
        if cache is not None and cache[0] is coords:
            feats = cache[1]
        elif self._heads_read_live_features():
            feats = self._interp(self.grid, coords)
        else:
            with torch.no_grad():                             # the heads detach it anyway: no autograd node, no saved tensors
                feats = self._interp(self.grid, coords)
        (sem_in, sem_grp), (inst_in, inst_grp) = self._head_inputs(feats, coords, channels)
        out = {}
        if "semantics" in channels and "inst_embedding" in channels and sem_in is inst_in and sem_grp is not None:
            # both heads read the same features: one autograd node, the input gradient is summed inside the kernels
            heads = ((*self._wide_head_weights(self.decoder_inst, sem_in), self.decoder_inst.input_dim),
                     (*self.decoder_semantics.weights(), self.decoder_semantics.input_dim))
            out["inst_embedding"], out["semantics"] = ops.head_composite_pair(sem_in, heads, w, alpha, ridx, pack_start, ray_of_pack, N,
                                                                              out_dtype=self.feat_dtype, x1_grouped=sem_grp)
            return out
        for ch, dec, x, grp in (("semantics", self.decoder_semantics, sem_in, sem_grp), ("inst_embedding", self.decoder_inst, inst_in, inst_grp)):
            if ch in channels:
                W, b = self._wide_head_weights(dec, x) if ch == "inst_embedding" else dec.weights()
                out[ch] = ops.head_composite(x, W, b, w, alpha, ridx, pack_start, ray_of_pack, N, in_dim=dec.input_dim,
                                             out_act=L.ACT_SOFTMAX, out_dtype=self.feat_dtype, x1_grouped=grp)
        return out

    # ---------------------------------------------------------------------------------------- prune
    @torch.no_grad()
    def prune(self, jitter=None):
        """Occupancy update (panoptic_delta_nef.py:63-104): EMA-max of the density at one jittered
        sample per dense cell, threshold (0.01*512)/sqrt(3), both grids get the new mask."""
        if self.grid is None:
            return
        density_decay = 0.6
        min_density = (0.01 * 512) / np.sqrt(3)
        dev = self.device
        g = self.grid
        points = g.dense_points.to(dev)
        res = 2.0 ** g.blas_level
        if jitter is None:
            jitter = torch.rand(points.shape[0], 3, device=dev)
        samples = (points.float() + jitter) / res * 2.0 - 1.0
        views = torch.zeros(points.shape[0], 3, device=dev)
        views[:, 2] = 1.0
        with g.fp32_coords():          # prune() runs outside the trainer's autocast region: fp32 coordinates (grids.rounds_coords)
            density = self.forward(coords=samples[:, None], ray_d=views, channels="density")
        g.occupancy = g.occupancy.to(dev).float().contiguous()
        bits = torch.empty(max(1, (g.num_cells + 31) // 32), dtype=torch.int32, device=dev)
        ops.occupancy_update(density.reshape(-1), g.occupancy, bits, density_decay, min_density)   # EMA-max + threshold + pack
        for grid in [self.grid] + ([self.delta_grid] if hasattr(self, "delta_grid") else []):
            grid.blas_init_bits(bits)


class PanopticNeF(PanopticDeltaNeF):
    """pc_nerf/panoptic_nef.py::PanopticNeF - the base field the delta class derives from in the reference, registrable under the same
    name (configs/bup20/lin_assign_app.yaml:71, lin_assign_direct_app.yaml:115, contrastive_delta_app.yaml:115): ONE grid; the semantic
    and the instance head read the main features (:338, :353), detached only when `sem_detach` / `inst_detach` say so - with a flag off
    the panoptic losses train the main grid as well; `inst_direct_pos` feeds the instance head the raw coordinates (:350-351).  Same
    kernels as the delta class: each head (+ its per-ray compositing in training) is one fused node whose input gradient flows back
    into the main grid's encode backward when its features are live.

    Quirks of the reference kept: `inst_direct_pos` is read by rgb_semantics (:350) but set by no constructor - a reference run has to
    assign the attribute (the YAML key reaches BaseNeuralField's **kwargs only); here it is also a constructor keyword (default False).
    With `inst_softmax` the instance head is `softmax(decoder(x))` whatever `inst_sigmoid` / `inst_normalize` say (:358 re-evaluates the
    decoder).  `panoptic_features_type` only sizes the heads' inputs (:99-107: 'position' -> 3): the semantic head is always fed the grid
    features (:338), so 'position' fails on the first semantic evaluation there (shape mismatch) and raises here; 'pos_encoding' reads an
    attribute the base class never defines (:103) and raises at construction, as there."""

    def __init__(self, *args, sem_detach=True, inst_detach=True, inst_direct_pos=False, panoptic_features_type=None, **kwargs):
        if panoptic_features_type == "pos_encoding":
            raise AttributeError("'PanopticNeF' object has no attribute 'pos_embed_dim'")         # panoptic_nef.py:103
        # the parent with 'appearance' / 'position' builds exactly this class' modules: one grid, heads sized for features / positions
        super().__init__(*args, sem_detach=sem_detach, inst_detach=inst_detach,
                         panoptic_features_type="position" if panoptic_features_type == "position" else "appearance", **kwargs)
        self.panoptic_features_type = panoptic_features_type
        self.sem_detach, self.inst_detach, self.inst_direct_pos = bool(sem_detach), bool(inst_detach), bool(inst_direct_pos)
        if self.inst_soft_temperature > 0.0:
            raise NotImplementedError("inst_soft_temperature is an option of PanopticDeltaNeF only")

    def get_nef_type(self):
        return "panoptic_nef"                                                                       # panoptic_nef.py:204-210

    def _pan_grouped(self):
        return self._grouped()

    def _heads_read_live_features(self):
        return not (self.sem_detach and (self.inst_detach or self.inst_direct_pos))       # :338, :353

    def _head_inputs(self, feats, coords, channels):
        grp = self._grouped()
        if "semantics" in channels and self.panoptic_features_type == "position":
            raise RuntimeError("mat1 and mat2 shapes cannot be multiplied: PanopticNeF feeds its semantic head the grid features "
                               "(panoptic_nef.py:338) but panoptic_features_type='position' sized it for 3 inputs (:99-101)")
        det = feats.detach()
        sem = det if self.sem_detach else feats                                                    # :338
        if self.inst_direct_pos:                                                                    # :350-351 (coordinates stay live)
            inst, inst_grp = self._pad8(coords.reshape(-1, 3).float(), self.feat_dtype), None
        else:
            inst = sem if self.inst_detach == self.sem_detach else (det if self.inst_detach else feats)   # :353
            inst_grp = grp
        return (sem, grp), (inst, inst_grp)

    def _inst_head(self, inst_in, inst_grp, mode):
        """:355-359 - with inst_softmax the output is softmax(decoder(x)): the sigmoid / normalize results are discarded."""
        if self.inst_softmax:
            return self.decoder_inst(inst_in, out_act=L.ACT_SOFTMAX, mode=mode, x1_grouped=inst_grp, out_dtype=self.feat_dtype)
        e = self.decoder_inst(inst_in, out_act=L.ACT_NONE, mode=mode, x1_grouped=inst_grp, out_dtype=self.feat_dtype)
        e = torch.sigmoid(e) if self.inst_sigmoid else e
        return F.normalize(e, dim=-1) if self.inst_normalize else e

    def rgb_semantics(self, coords, ray_d=None, compute_channels=None, pidx=None, lod_idx=None, ridx=None, ray_dirs=None,
                      ray_packs=None):
        out = super().rgb_semantics(coords, ray_d=ray_d, compute_channels=compute_channels, pidx=pidx, lod_idx=lod_idx, ridx=ridx,
                                    ray_dirs=ray_dirs, ray_packs=ray_packs)
        if self.inst_direct_pos and "inst_embedding" in out:                                        # decoder_inst(coords): [batch, num_samples, I]
            out["inst_embedding"] = out["inst_embedding"].reshape(coords.shape[0], coords.shape[1], -1)
        return out

    def can_fuse_panoptic(self, channels):
        if self.precision != "bf16" or self._grouped() is None:
            return False
        ok = True
        if "semantics" in channels:
            ok &= self.sem_softmax and not (self.sem_sigmoid or self.sem_normalize) and self.panoptic_features_type != "position"
        if "inst_embedding" in channels:
            ok &= self.inst_softmax and not self.inst_direct_pos
        return bool(ok)
