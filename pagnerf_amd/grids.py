"""Feature grids with the kaolin-wisp grid plugin API, backed by the gfx950 encode kernels.

  HashGridHIP     replaces grids/hash_grid_torch.py::HashGridTorch (+ HashEmbedder) and, with a
                  geometric resolution list, grids/hash_grid_tinycudann.py::HashGridTinyCudaNN
  PermutoGridHIP  replaces grids/permuto_grid.py::PermutoGrid (+ permutohedral_encoding.PermutoEncoding)

Both keep the contract the nef/tracer rely on: ctor swallows **kwargs,
`interpolate(coords[B,S,3], lod_idx, pidx=None) -> [B*S, L*F]`, `raymarch(rays, level, num_samples,
raymarch_type)`, attributes num_lods / active_lods / feature_dim / multiscale_type / blas_level /
dense_points / occupancy, `copy.deepcopy`-able, tables + occupancy saved through state_dict.
The octree BLAS of wisp (third party) is replaced by a dense occupancy bitfield at 2^blas_level.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import ops


class OccupancyBLAS(nn.Module):
    """Dense occupancy bitfield standing in for wisp's OctreeAS (grids/occtree.py:54-67 is the
    in-tree description of that contract).  Bit (x*R + y)*R + z of `blas_bits`."""

    def __init__(self, blas_level=7):
        super().__init__()
        self.blas_level = int(blas_level)
        R = 2 ** self.blas_level
        self.num_cells = R ** 3
        words = max(1, self.num_cells // 32)
        self.register_buffer("blas_bits", torch.full((words,), -1, dtype=torch.int32))
        self.occupancy = torch.zeros(self.num_cells)
        self._dense_points = None
        self._all_occupied = True
        self._pack_cache = None
        self._coarse = None          # (blas_bits identity, version) -> coarse bitfield of the voxel march (ops.occupancy_coarse)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        """A checkpoint replaces buffers the kernels do not read directly: the march takes `_all_occupied`, the encoders a host copy of
        the per-level shifts (`_spec`).  Both are derived state - rebuild them from what was just loaded (resume pattern of the
        reference: main_hp_tunning.py:197 `pipeline.load_state_dict`)."""
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
        self._refresh_derived()

    def _refresh_derived(self):
        self._all_occupied = bool((self.blas_bits == -1).all()) if self.num_cells >= 32 else False
        self._pack_cache = None
        self._coarse = None

    def _coarse_bits(self, bits):
        """Coarse occupancy for the voxel march, rebuilt when the bitfield object or its contents (version counter) change."""
        key = (id(bits), bits._version, str(bits.device))
        if self._coarse is None or self._coarse[0] != key:
            self._coarse = (key, ops.occupancy_coarse(bits, self.blas_level))
        return self._coarse[1]

    @property
    def dense_points(self):
        """int [R^3,3] cell coordinates, x slowest (same linear order as the bitfield)."""
        if self._dense_points is None:
            R = 2 ** self.blas_level
            ar = torch.arange(R, dtype=torch.int16)
            self._dense_points = torch.stack(torch.meshgrid(ar, ar, ar, indexing="ij"), -1).reshape(-1, 3)
        return self._dense_points

    def blas_init(self, mask):
        """Rebuild from a bool [R^3] mask of kept cells (panoptic_delta_nef.py:98-104)."""
        mask = mask.reshape(-1).to(torch.bool)
        assert mask.numel() == self.num_cells
        pad = (-mask.numel()) % 32
        if pad:
            mask = torch.cat([mask, mask.new_zeros(pad)])
        w = (mask.reshape(-1, 32).long() << torch.arange(32, device=mask.device)).sum(1)
        w = torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32)
        self.blas_bits = w.to(self.blas_bits.device)
        self._refresh_derived()              # also drops the coarse-occupancy and pack caches keyed on the old tensor

    def blas_init_bits(self, bits):
        """Adopt a packed bitfield produced on the device (ops.occupancy_update)."""
        assert bits.dtype == torch.int32 and bits.numel() == self.blas_bits.numel()
        self.blas_bits = bits.clone()
        self._refresh_derived()

    def occupancy_mask(self):
        bits = self.blas_bits.long() & 0xFFFFFFFF
        m = ((bits[:, None] >> torch.arange(32, device=bits.device)) & 1).bool().reshape(-1)
        return m[:self.num_cells]

    def raymarch_voxel_begin(self, rays, num_samples, max_travel=None):
        """The voxel march's walk, queued (ops.raymarch_voxel_begin); raymarch_voxel_finish() sizes and fills the packed tensors.  raymarch(..., 'voxel') = both."""
        bits = None if self._all_occupied else self.blas_bits
        if bits is not None and bits.device != rays.origins.device:
            self.blas_bits = bits = bits.to(rays.origins.device)
        coarse = self._coarse_bits(bits) if bits is not None else None
        return ops.raymarch_voxel_begin(rays.origins, rays.dirs, rays.dist_min, rays.dist_max, num_samples, bits, self.blas_level, max_travel=max_travel,
                                        occupancy_coarse_bits=coarse, want_packs=True)

    def raymarch_voxel_finish(self, state, rays, num_samples):
        ridx, pidx, samples, depths, deltas, boundary, pack_start, ray_of_pack, ridx_sample, ridx64 = ops.raymarch_voxel_finish(state)
        self._pack_cache = (ridx64, ridx_sample, pack_start, ray_of_pack)
        if torch.is_grad_enabled() and (rays.origins.requires_grad or rays.dirs.requires_grad) and ridx.numel():
            # pose gradient: the k samples of a ray's nuggets are its pack (pack_start counts samples) - the same per-ray segmented
            # sums as in 'ray' mode (pag_ray_sample_grad: one launch, fixed order)
            samples = ops.ray_samples(rays.origins, rays.dirs, samples, depths, pack_start, ray_of_pack, ridx=ridx_sample)
        return ridx64, pidx, samples, depths[..., None], deltas[:, None], boundary

    accepts_max_travel = True      # raymarch(..., max_travel=) applies the tracer's travel filter inside the voxel walk

    def raymarch(self, rays, level=None, num_samples=64, raymarch_type="ray", jitter=None, max_travel=None):
        """'ray'  : (ridx i64[M], pidx i32[M], samples [M,1,3], depths [M,1], deltas [M,1], boundary bool[M])
        'voxel': per nugget ridx i64[M'], pidx i32[M']; samples [M',k,3], depths [M',k,1], deltas [M'*k,1], boundary bool[M'*k]
        (the shapes tracers/panoptic_packed_rf_tracer.py:88-108 indexes; k = num_samples).  max_travel ('voxel' only): the
        travel filter of tracer :88-108 already applied - the tracer then skips its own."""
        bits = None if self._all_occupied else self.blas_bits
        if bits is not None and bits.device != rays.origins.device:
            self.blas_bits = bits = bits.to(rays.origins.device)
        if raymarch_type == "voxel":
            return self.raymarch_voxel_finish(self.raymarch_voxel_begin(rays, num_samples, max_travel), rays, num_samples)
        if raymarch_type != "ray":
            raise NotImplementedError("raymarch_type '%s'" % raymarch_type)
        ridx, pidx, samples, depths, deltas, boundary, pack_start, ray_of_pack, ridx64 = ops.raymarch_ray(
            rays.origins, rays.dirs, rays.dist_min, rays.dist_max, num_samples, jitter, bits, self.blas_level, want_ridx64=True)
        self._pack_cache = (ridx64, ridx, pack_start, ray_of_pack)
        if torch.is_grad_enabled() and (rays.origins.requires_grad or rays.dirs.requires_grad) and ridx.numel():
            samples = ops.ray_samples(rays.origins, rays.dirs, samples, depths, pack_start, ray_of_pack, ridx=ridx)   # pose gradient
        out = samples[:, None]
        if hasattr(samples, "_pag_rays"):
            out._pag_rays = samples._pag_rays          # the tag travels with the [M,1,3] view the tracer hands to the nef
        return ridx64, pidx, out, depths[:, None], deltas[:, None], boundary


class _GridBase(OccupancyBLAS):
    def __init__(self, feature_dim, base_lod=2, num_lods=1, interpolation_type="linear", multiscale_type="cat",
                 feature_std=0.0, feature_bias=0.0, blas_level=7, table_dtype=torch.float32, **kwargs):
        super().__init__(blas_level)
        self.feature_dim = int(feature_dim)
        self.base_lod = base_lod
        self.num_lods = int(num_lods)
        self.interpolation_type = interpolation_type
        self.multiscale_type = multiscale_type
        self.feature_std, self.feature_bias = feature_std, feature_bias
        self.table_dtype = table_dtype
        self.active_lods = list(range(self.num_lods))
        self.max_lod = self.num_lods - 1
        self.tables = None
        self._spec = None

    def _finish(self, feats, batch, num_samples):
        if self.multiscale_type == "cat":
            return feats
        if self.multiscale_type == "sum":
            return feats.reshape(batch, num_samples, self.num_lods, self.feature_dim).sum(-2)
        raise NotImplementedError(self.multiscale_type)

    def _coords(self, coords):
        return coords.reshape(-1, 3)

    def rounds_coords(self):
        """Whether interpolate() rounds the coordinates to fp16 in THIS call.  The reference's `custom_fwd(cast_inputs=torch.half)`
        (grids/permuto_grid.py:65, grids/hash_grid_tinycudann.py:36) acts only inside the train step's `torch.cuda.amp.autocast()`
        (pc_nerf/trainer.py:429); validate() / evaluate_metrics() (pipeline.eval(), no autocast: trainer.py:630,944) and nef.prune()
        see fp32 coordinates.  half_coords=True follows that: rounding under autocast, or in training mode outside fp32_coords()
        (training mode stands for "inside the train step" for callers that do not open an autocast region); "always" / False
        force it on / off."""
        hc = self.half_coords
        if hc == "always":
            return True
        if not hc:
            return False
        return torch.is_autocast_enabled() or (self.training and not getattr(self, "_fp32_coords", False))

    def fp32_coords(self):
        """Context manager: interpolate() keeps fp32 coordinates even in training mode (nef.prune(), panoptic_delta_nef.py:63-104,
        runs outside the trainer's autocast region)."""
        grid = self

        class _Ctx:
            def __enter__(self):
                self.prev, grid._fp32_coords = getattr(grid, "_fp32_coords", False), True

            def __exit__(self, *exc):
                grid._fp32_coords = self.prev
        return _Ctx()

    def interpolate_scaled(self, coords, feat_scale=None, out_dtype=torch.float32, layout=None, addend=None):
        """interpolate() with the nef's lod_weights folded into the kernel; layout="xcd8" returns the bf16
        [8, M, 8] XCD-grouped features the fused decoders consume (ops.encode); addend: see ops.encode."""
        return ops.encode(self._coords(coords), self.tables, self._spec, feat_scale, out_dtype, False, layout=layout, addend=addend,
                          half_coords=self.rounds_coords(), rays=getattr(coords, "_pag_rays", None))


class HashGridHIP(_GridBase):
    """Multiresolution hash grid (every level hashed, fp32-derived resolutions - Appendix E.7)."""

    def __init__(self, feature_dim, codebook_bitwidth=19, half_coords=False, **kwargs):
        super().__init__(feature_dim, **kwargs)
        self.codebook_bitwidth = int(codebook_bitwidth)
        # grids/hash_grid_tinycudann.py:36 rounds the coordinates to fp16 under autocast; grids/hash_grid_torch.py does not (Appendix E.6)
        self.half_coords = half_coords if half_coords == "always" else bool(half_coords)

    @staticmethod
    def level_resolutions(base_resolution, finest_resolution, n_levels):
        """grids/hash_grid_torch.py:59,99 in fp32 (16..2048 over 16 levels ends at 2047)."""
        base, fine = torch.tensor(base_resolution), torch.tensor(finest_resolution)
        if n_levels == 1:
            return [float(base)]
        b = torch.exp((torch.log(fine) - torch.log(base)) / (n_levels - 1))
        return [float(torch.floor(base * b ** i)) for i in range(n_levels)]

    def init_from_resolutions(self, resolutions, exact=False):
        """As the reference: only resolutions[0], [-1] and len are used (hash_grid_torch.py:126-128)
        unless exact=True (tinycudann-style explicit list)."""
        self.resolutions = list(resolutions)
        self.num_lods = len(resolutions)
        self.active_lods = list(range(self.num_lods))
        self.max_lod = self.num_lods - 1
        res = [float(r) for r in resolutions] if exact else self.level_resolutions(resolutions[0], resolutions[-1], self.num_lods)
        self.level_res = res
        T = 2 ** self.codebook_bitwidth
        dev = self.blas_bits.device
        t = torch.empty(self.num_lods, T, self.feature_dim, device=dev).uniform_(-1e-4, 1e-4)   # hash_grid_torch.py:65
        self.tables = nn.Parameter(t.to(self.table_dtype))
        self._spec = ops.hash_spec(res, self.codebook_bitwidth, self.feature_dim, half_coords=bool(self.half_coords))

    def init_from_geometric(self, min_width, max_width, num_lods):
        """wisp HashGrid.init_from_geometric (config_parser.py:733): int(1 + floor(min * b**l))."""
        b = math.exp((math.log(max_width) - math.log(min_width)) / (num_lods - 1))
        self.init_from_resolutions([int(1 + math.floor(min_width * b ** l)) for l in range(num_lods)])

    def interpolate(self, coords, lod_idx=None, pidx=None):
        batch, num_samples, _ = coords.shape
        if coords.numel() == 0:
            return torch.empty(0, self.num_lods * self.feature_dim, device=coords.device)
        return self._finish(self.interpolate_scaled(coords), batch, num_samples)


class PermutoGridHIP(_GridBase):
    """Permutohedral-lattice hash grid (grids/permuto_grid.py)."""

    def __init__(self, feature_dim, coarsest_scale=1.0, finest_scale=0.001, capacity_log_2=18, num_lods=24,
                 half_coords=True, **kwargs):
        kwargs.pop("multiscale_type", None)
        super().__init__(feature_dim, num_lods=num_lods, multiscale_type="cat", **kwargs)   # permuto_grid.py:31
        self.coarsest_scale, self.finest_scale = coarsest_scale, finest_scale
        self.capacity = 2 ** int(capacity_log_2)
        # permuto_grid.py:65,71 - under the trainer's autocast the coordinates are rounded to fp16 before the encoder sees them
        # (PAG_ENC_HALF_COORDS: done inside the kernels).  On by default because that is how the reference trains.
        self.half_coords = half_coords if half_coords == "always" else bool(half_coords)

    def set_capacity(self, capacity_log_2):
        self.capacity = 2 ** int(capacity_log_2)

    @staticmethod
    def scale_factors(scales):
        scales = np.asarray(scales, dtype=np.float64)
        sf = np.stack([1.0 / (np.sqrt((i + 1) * (i + 2)) * scales) for i in range(3)], 1)
        return torch.from_numpy(sf.astype(np.float32))

    def init_from_scales(self, random_shift=None, tables=None):
        self.active_lods = list(range(self.num_lods))
        self.max_lod = self.num_lods - 1
        self.resolutions = np.geomspace(self.coarsest_scale, self.finest_scale, num=self.num_lods)   # permuto_grid.py:53
        dev = self.blas_bits.device
        if random_shift is None:
            random_shift = torch.randn(self.num_lods, 3) * 10.0
        self.register_buffer("random_shift_per_level", random_shift.float().cpu().clone())
        if tables is None:
            tables = torch.randn(self.num_lods, self.capacity, self.feature_dim, device=dev) * 1e-5
        self.tables = nn.Parameter(tables.to(device=dev, dtype=self.table_dtype))
        self._build_spec()

    def _build_spec(self):
        self._spec = ops.permuto_spec(self.scale_factors(self.resolutions), self.random_shift_per_level, self.capacity,
                                      self.feature_dim, half_coords=bool(self.half_coords))

    def _refresh_derived(self):
        super()._refresh_derived()
        if getattr(self, "_spec", None) is not None and "random_shift_per_level" in self._buffers:
            self._build_spec()        # the kernels hash with the host copy of the (just loaded) per-level shifts

    def interpolate(self, coords, lod_idx=None, pidx=None):
        if coords.numel() == 0:
            return torch.empty([0, 1, self.num_lods * self.feature_dim], device=coords.device)   # permuto_grid.py:68-69
        return self.interpolate_scaled(coords)
