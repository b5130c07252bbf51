"""Minimal stand-ins for the wisp.core data classes the hot path exchanges with its callers
(wisp itself is a third-party dependency of the reference and is not part of this build).

Uses in the reference that define the contract:
  Rays          pc_nerf/trainer.py:422,644,958 ; datasets/transforms/ray_sampler.py:19-38 ;
                pc_nerf/ba_pipeline.py:88-92
  RenderBuffer  tracers/panoptic_packed_rf_tracer.py:195 ; pc_nerf/trainer.py:648,706,710
  Pipeline      pc_nerf/ba_pipeline.py:73-76 (forward = tracer(nef, ...) or nef(...))
"""
import torch
import torch.nn as nn


class Rays:
    def __init__(self, origins, dirs, dist_min=0.0, dist_max=6.0):
        self.origins, self.dirs, self.dist_min, self.dist_max = origins, dirs, dist_min, dist_max

    @property
    def shape(self):
        return self.origins.shape[:-1]

    def __len__(self):
        return self.origins.shape[0]

    def _map(self, fn):
        return Rays(fn(self.origins), fn(self.dirs), self.dist_min, self.dist_max)

    def reshape(self, *dims):
        return self._map(lambda t: t.reshape(*dims))

    def to(self, *a, **k):
        return self._map(lambda t: t.to(*a, **k))

    def __getitem__(self, idx):
        return self._map(lambda t: t[idx])

    def split(self, n):
        return [Rays(o, d, self.dist_min, self.dist_max) for o, d in zip(self.origins.split(n), self.dirs.split(n))]

    @classmethod
    def cat(cls, rays_list, dim=0):
        return cls(torch.cat([r.origins for r in rays_list], dim), torch.cat([r.dirs for r in rays_list], dim),
                   rays_list[0].dist_min, rays_list[0].dist_max)

    @classmethod
    def stack(cls, rays_list, dim=0):
        return cls(torch.stack([r.origins for r in rays_list], dim), torch.stack([r.dirs for r in rays_list], dim),
                   rays_list[0].dist_min, rays_list[0].dist_max)


class RenderBuffer:
    """Named per-ray channels; `a += b` concatenates along the ray axis (trainer.py:648).
    Channels are plain instance attributes, as in wisp's dataclass: the reference trainer finds them by introspection
    (`'ray_sparcity_loss' in dir(rb)` trainer.py:438, `vars(rb)['inst_embedding']` trainer.py:488)."""

    def __init__(self, **channels):
        self.__dict__.update(channels)

    @property
    def channels(self):
        return set(self.__dict__)

    def _items(self):
        return list(self.__dict__.items())

    def _map(self, fn):
        return RenderBuffer(**{k: (fn(v) if isinstance(v, torch.Tensor) and v.dim() > 0 else v) for k, v in self._items()})

    def __iadd__(self, other):
        for k, v in other._items():
            mine = self.__dict__.get(k)
            if isinstance(v, torch.Tensor) and v.dim() > 0 and mine is not None:
                self.__dict__[k] = torch.cat([mine, v], 0)
            else:
                self.__dict__[k] = v if mine is None else mine
        return self

    def reshape(self, *dims):
        return self._map(lambda t: t.reshape(*dims))

    def cpu(self):
        return self._map(lambda t: t.cpu())

    def detach(self):
        return self._map(lambda t: t.detach())

    def to(self, *a, **k):
        return self._map(lambda t: t.to(*a, **k))

    def byte(self):
        return self._map(lambda t: (t.clamp(0, 1) * 255).byte() if t.is_floating_point() else t)

    def image(self):
        return self


class Pipeline(nn.Module):
    def __init__(self, nef, tracer=None):
        super().__init__()
        self.nef, self.tracer = nef, tracer

    def forward(self, *args, **kwargs):
        if self.tracer is not None:
            return self.tracer(self.nef, *args, **kwargs)
        return self.nef(*args, **kwargs)


def batch_render(pipeline, rays, channels=("rgb",), render_batch=4000, cam_ids=None):
    """Validation-time chunked render (pc_nerf/trainer.py:637-649): BAPipelines first map the base rays to world space,
    then the rays go through the pipeline `render_batch` at a time and the per-chunk RenderBuffers are joined along the ray axis -
    the result of the reference's `rb += render(ray_pack)` loop, but with ONE concatenation per channel at the end: `+=` re-copies
    everything rendered so far for every chunk (116 chunks of a 720 x 1280 image: ~45 GB of copies for 0.8 GB of output)."""
    if hasattr(pipeline, "transform_rays") and cam_ids is not None:
        rays = pipeline.transform_rays(rays, cam_ids)
    tracer = getattr(pipeline, "tracer", None)
    if tracer is not None and hasattr(tracer, "render_packs") and not torch.is_grad_enabled():
        # this package's tracer marches pack i + 1 on a second stream while pack i is shaded (same launches, same values: PanopticPackedRFTracer.render_packs)
        parts = tracer.render_packs(pipeline.nef, rays.split(render_batch), channels=channels, lod_idx=None)
    else:
        parts = [pipeline(rays=pack, lod_idx=None, channels=channels) for pack in rays.split(render_batch)]
    if not parts:
        return None
    if len(parts) == 1:
        return parts[0]
    out = {}
    for part in parts:                     # channels in first-seen order over ALL chunks: one that is None in the first chunk and a tensor later is kept, as `+=` keeps it
        for k, v in part._items():
            if k in out and out[k] is not None:
                continue
            if isinstance(v, torch.Tensor) and v.dim() > 0:
                out[k] = torch.cat([p.__dict__[k] for p in parts if isinstance(p.__dict__.get(k), torch.Tensor) and p.__dict__[k].dim() > 0], 0)
            else:
                out[k] = v                 # scalars (e.g. a regularisation loss) keep the first chunk's value, as `+=` does
    return RenderBuffer(**out)
