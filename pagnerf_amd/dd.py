"""Delta-density variant: same kernels as the main path plus one extra density stream.

  PanopticDDensityNeF             <- pc_nerf/panoptic_dd_nef.py::PanopticDDensityNeF
  PanopticDDensityPackedRFTracer  <- tracers/panoptic_dd_packed_rf_tracer.py::PanopticDDensityPackedRFTracer

What differs from PanopticDeltaNeF / PanopticPackedRFTracer:
  * a fifth decoder `decoder_delta_density` (features -> 1, activation 'none', panoptic_dd_nef.py:49-56) on the panoptic
    features; `panoptic_density = relu(density_feats[...,0:1].detach() + delta_density)` (:243-247);
  * the tracer composites semantics / instances with the weights and alpha of the PANOPTIC density, which keep their
    gradient (only deltas / boundary are detached, tracer :124-135,:162-166) - ops.composite_features;
  * prune() keeps a cell when either grid's EMA-max occupancy passes the threshold (:63-117).
A decoder without activations is one affine map, so decoder_delta_density is evaluated as a single composed row vector
(lout.weight @ layers[0].weight); autograd through the composition delivers the gradients of both nn.Linear modules.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as L
from . import ops
from .core import RenderBuffer
from .nef import PanopticDeltaNeF
from .tracer import PanopticPackedRFTracer


class LinearChainDecoder(nn.Module):
    """wisp BasicDecoder with activation 'none': parameter layout (layers[i], lout) kept, evaluated as one affine map."""

    def __init__(self, input_dim, output_dim, num_layers=1, hidden_dim=64):
        super().__init__()
        dims = [input_dim] + [hidden_dim] * num_layers
        self.layers = nn.ModuleList([nn.Linear(dims[i], dims[i + 1]) for i in range(num_layers)])
        self.lout = nn.Linear(dims[-1], output_dim)
        self.input_dim, self.output_dim = input_dim, output_dim

    def affine(self):
        W, b = self.lout.weight, self.lout.bias
        for lin in reversed(self.layers):
            b = b + W @ lin.bias
            W = W @ lin.weight
        return W, b                                            # [out, in], [out]

    def forward(self, x, grouped=None):
        """x [M,in] or the encoders' bf16 [8,M,8] XCD-grouped tensor (grouped = (levels, feats)) -> f32 [M,out]."""
        W, b = self.affine()
        if grouped is None:
            return F.linear(x.float(), W, b)
        return ops.affine_xcd8(x, W, b, grouped)          # one launch on the bf16 [8, M, 8] features (pag_affine_xcd8_fwd)


class PanopticDDensityNeF(PanopticDeltaNeF):
    def __init__(self, delta_num_layers=1, delta_hidden_dim=64, separate_sem_grid=False, inst_soft_temperature=0.0, **kwargs):
        kwargs.setdefault("panoptic_features_type", "separate" if separate_sem_grid else "delta")
        super().__init__(inst_soft_temperature=inst_soft_temperature, delta_num_layers=delta_num_layers,
                         delta_hidden_dim=delta_hidden_dim, **kwargs)
        self.separate_sem_grid = separate_sem_grid
        eff = self.feature_dim * self.num_lods
        if delta_num_layers == 0:
            delta_hidden_dim = eff                                                           # panoptic_dd_nef.py:47-48
        self.decoder_delta_density = LinearChainDecoder(eff, 1, delta_num_layers, delta_hidden_dim)
        self._fns = [(self.rgb_semantics, {"density", "rgb", "delta_density", "panoptic_density", "semantics", "inst_embedding"})]

    def rgb_semantics(self, coords, ray_d=None, compute_channels=None, pidx=None, lod_idx=None, ridx=None, ray_dirs=None,
                      ray_packs=None):
        if isinstance(compute_channels, str):
            compute_channels = {compute_channels}
        extra = {"delta_density", "panoptic_density"} & set(compute_channels or ())
        base = set(compute_channels or ()) - extra
        if extra:
            base |= {"density"}
        out = super().rgb_semantics(coords, ray_d=ray_d, compute_channels=base, pidx=pidx, lod_idx=lod_idx, ridx=ridx,
                                    ray_dirs=ray_dirs, ray_packs=ray_packs) if base else {}
        if extra:
            batch, num_samples, _ = coords.shape
            feats = self._feat_cache[1].detach()
            delta = self._interp(self.delta_grid, coords.detach())
            pan = delta if self.separate_sem_grid else feats + delta                         # :231-234
            dd = self.decoder_delta_density(pan, self._grouped()).reshape(batch, num_samples, 1)   # :238
            if "delta_density" in extra:
                out["delta_density"] = dd
            if "panoptic_density" in extra:                                                  # :243-247
                pre = self._density_feats.detach()[:, 0:1].float().reshape(batch, num_samples, 1)
                out["panoptic_density"] = torch.relu(dd if self.separate_sem_grid else pre + dd)
        if compute_channels is not None and "density" not in compute_channels:
            out.pop("density", None)
        return out

    @torch.no_grad()
    def prune(self, jitter=None):
        """:63-117 - both grids keep their own EMA-max occupancy; a cell survives when either passes the threshold."""
        if self.grid is None:
            return
        density_decay, min_density = 0.6, (0.01 * 512) / np.sqrt(3)
        dev = self.device
        g = self.grid
        points = g.dense_points.to(dev)
        res = 2.0 ** g.blas_level
        if jitter is None:
            jitter = torch.rand(points.shape[0], 3, device=dev)
        samples = (points.float() + jitter) / res * 2.0 - 1.0
        views = torch.zeros(points.shape[0], 3, device=dev)
        views[:, 2] = 1.0
        words = max(1, (g.num_cells + 31) // 32)
        keep = torch.zeros(words, dtype=torch.int32, device=dev)
        for grid, channel in ((self.grid, "density"), (self.delta_grid, "panoptic_density")):
            with self.grid.fp32_coords(), self.delta_grid.fp32_coords():      # outside the trainer's autocast region (grids.rounds_coords)
                density = self.forward(coords=samples[:, None], ray_d=views, channels=channel)
            grid.occupancy = grid.occupancy.to(dev).float().contiguous()
            bits = torch.empty(words, dtype=torch.int32, device=dev)
            ops.occupancy_update(density.reshape(-1).float(), grid.occupancy, bits, density_decay, min_density)
            keep |= bits
        for grid in (self.grid, self.delta_grid):
            grid.blas_init_bits(keep)


class PanopticDDensityPackedRFTracer(PanopticPackedRFTracer):
    def __init__(self, ray_sparcity_reg=0.0, **kwargs):
        super().__init__(ray_sparcity_reg=ray_sparcity_reg, **kwargs)
        self.panoptic_channels = {"delta_density", "panoptic_density", "semantics", "inst_embedding"}

    def get_supported_channels(self):
        return {"depth", "hit", "rgb", "alpha", "delta_density", "panoptic_density", "semantics", "inst_embedding"}

    def get_required_nef_channels(self):
        return {"rgb", "density", "panoptic_density"}

    def trace(self, nef, channels, extra_channels, rays, lod_idx=None, raymarch_type="voxel", num_steps=64, step_size=1.0,
              bg_color="white", stage="val", jitter=None):
        assert nef.grid is not None, "this tracer requires a grid"                            # :79
        N = rays.origins.shape[0]
        dev = rays.origins.device
        if lod_idx is None:
            lod_idx = nef.grid.num_lods - 1
        kw = {"jitter": jitter} if jitter is not None else {}
        ridx, pidx, samples, depths, deltas, boundary = nef.grid.raymarch(                     # :88-89
            rays, level=nef.grid.active_lods[lod_idx], num_samples=num_steps, raymarch_type=raymarch_type, **kw)
        k = samples.shape[1] if samples.dim() == 3 else 1
        cache = getattr(nef.grid, "_pack_cache", None)
        if cache is not None and cache[0] is ridx:
            _, ridx32, pack_start, ray_of_pack = cache
        else:
            ridx32 = ridx.int() if k == 1 else ridx.int().repeat_interleave(k)
            pack_start, ray_of_pack = ops.packs_from_boundary(ridx32, boundary)
        outputs = {}
        sample_channels = set(channels - self.render_channels)                                 # :101-105
        sample_channels.update(["density"])
        pan_req = [c for c in channels if c in self.panoptic_channels]
        if pan_req:
            sample_channels.update(["panoptic_density"])
        if getattr(nef, "accepts_ray_index", False):
            feats = nef(coords=samples, ridx=ridx32, ray_dirs=rays.dirs, pidx=pidx, lod_idx=lod_idx, channels=sample_channels,
                        ray_packs=(pack_start, ray_of_pack))
        else:
            feats = nef(coords=samples, ray_d=rays.dirs.index_select(0, ridx), pidx=pidx, lod_idx=lod_idx, channels=sample_channels)
        sigma = feats["density"].reshape(-1)
        if self.ray_sparcity_reg > 0.0 and stage == "train":                                   # :108-111
            per = torch.log(1.0 + 2 * sigma ** 2)
            outputs["ray_sparcity_loss"] = torch.zeros(N, device=dev).scatter_add(0, ridx32.long(), per).mean() * self.ray_sparcity_reg
        rgb = feats["rgb"].reshape(-1, 3) if "rgb" in channels else None
        dep = depths.reshape(-1) if "depth" in channels else None
        alpha, hit, out_rgb, out_depth, w = ops.composite(sigma, rgb, deltas.reshape(-1), dep, pack_start, ray_of_pack, N,
                                                          bg_white=(bg_color == "white"))       # :114-160
        outputs["alpha"] = alpha[:, None]
        outputs["hit"] = hit.view(torch.bool)          # the kernel writes 0 / 1 bytes: reinterpret, no cast launch
        if rgb is not None:
            outputs["rgb"] = out_rgb
        if dep is not None:
            outputs["depth"] = out_depth[:, None]
        if pan_req:                                                                             # :124-135, :162-166
            psigma = feats["panoptic_density"].reshape(-1)
            for ch in pan_req:
                f = feats[ch].reshape(-1, feats[ch].shape[-1])
                outputs[ch], _ = ops.composite_features(psigma, deltas.reshape(-1), f, ridx32, pack_start, ray_of_pack, N)
        extra_outputs = {}
        for ch in extra_channels:                                                               # :168-176
            f = nef(coords=samples, ray_d=rays.dirs.index_select(0, ridx), pidx=pidx, lod_idx=lod_idx, channels=ch)
            extra_outputs[ch] = ops.composite_feats(f.reshape(-1, f.shape[-1]), w, alpha.detach(), pack_start, ray_of_pack, N)
        return RenderBuffer(**outputs, **extra_outputs)
