"""PanopticPackedRFTracer on the gfx950 kernels - same channels, arguments and output buffers as
tracers/panoptic_packed_rf_tracer.py::PanopticPackedRFTracer.

trace() follows the reference step by step (file:line of the reference in comments); the kaolin
calls it makes (mark_pack_boundaries / exponential_integration / sum_reduce, :114,:135,:138,...)
are replaced by two fused launches: ops.composite (segmented scan + alpha/rgb/depth sums) and
ops.composite_feats (wide per-ray feature sums).  Quirks kept on purpose:
colour and panoptic features are multiplied by alpha a second time, depth is not; panoptic
channels use weights that carry no gradient; rays without samples keep the background.
"""
import inspect

import torch
import torch.nn as nn

from . import ops
from .core import RenderBuffer


_TRACE_PARAMS = {}      # per tracer class: trace()'s parameters (inspect.signature is ~40 us per call)


class PanopticPackedRFTracer(nn.Module):
    def __init__(self, ray_sparcity_reg=0.0, ray_max_travel=6.0, raymarch_type="voxel", num_steps=64, step_size=1.0,
                 bg_color="white", use_graphs=None, graph_split=None, **kwargs):
        """use_graphs (this build's addition; default: the PAG_GRAPHS environment variable, else ON since round 5): training-time traces
        (`stage='train'` with gradients enabled, as pc_nerf/trainer.py:435 calls the pipeline) replay the post-march part of the step - forward
        and backward - as HIP graphs over static, padded sample buffers (pagnerf_amd/graphs.py).  Same channels, same values; what changes is
        that the host neither waits for the sample count nor issues ~35 launches per step (the 0.5 - 3 ms post-prune steps, three quarters of a
        best.yaml run, are host-bound without it).  Validation / no_grad traces, extra channels and foreign nefs / grids take the eager path by
        themselves; `use_graphs=False` (or PAG_GRAPHS=0) switches the graphs off, "static" keeps the static buffers without the capture.
        Ownership rules of the graph path: INTEGRATION.md section 6."""
        super().__init__()
        import os
        if use_graphs is None:
            use_graphs = os.environ.get("PAG_GRAPHS", "1")
            use_graphs = "static" if use_graphs == "static" else bool(int(use_graphs))
        # True: HIP graphs; "static": the graph path's static padded buffers and optimistic sample-count check with eager launches (no capture)
        self.use_graphs = "static" if use_graphs == "static" else bool(use_graphs)
        # graph_split: capture the backward as two graphs (panoptic heads | the rest) so that gradient hooks fire between them; None = only
        # when more than one rank trains (pagnerf_amd/graphs.py::_Graphed)
        self.graph_split = graph_split
        self._graphs = None
        self.raymarch_type, self.num_steps, self.step_size, self.bg_color = raymarch_type, num_steps, step_size, bg_color
        self.render_channels = {"depth", "alpha", "hit"}
        self.base_channels = {"rgb", "density"}
        self.panoptic_channels = {"semantics", "inst_embedding"}
        self.ray_sparcity_reg = ray_sparcity_reg
        self.ray_max_travel = ray_max_travel

    def get_supported_channels(self):
        return {"depth", "hit", "rgb", "alpha", "semantics", "inst_embedding"}

    def get_required_nef_channels(self):
        return {"rgb", "density"}

    def forward(self, nef, channels=None, extra_channels=None, **kwargs):
        """wisp BaseTracer.forward: trace() arguments come from kwargs, else
        from a same-named attribute of the tracer, else from the Python default."""
        requested, extra, args = self._resolve(nef, channels, extra_channels, kwargs)
        return self.trace(nef, requested - extra, extra, **args)

    def _resolve(self, nef, channels, extra_channels, kwargs, skip=()):
        """wisp BaseTracer.forward's bookkeeping: -> (requested channels, extra channels, trace() arguments from kwargs / attributes / defaults)."""
        missing = self.get_required_nef_channels() - nef.get_supported_channels()
        if missing:
            raise Exception("nef does not supply the channels this tracer needs: %s" % missing)
        if channels is None:
            requested = self.get_supported_channels()
        elif isinstance(channels, str):
            requested = {channels}
        else:
            requested = set(channels)
        extra = requested - self.get_supported_channels()
        if extra_channels is not None:
            extra |= {extra_channels} if isinstance(extra_channels, str) else set(extra_channels)
        unsupported = extra - nef.get_supported_channels()
        if unsupported:
            raise Exception("channels %s are supported by neither the tracer nor the nef" % unsupported)
        args = {}
        sig = _TRACE_PARAMS.get(type(self))
        if sig is None:
            sig = _TRACE_PARAMS[type(self)] = tuple(inspect.signature(self.trace).parameters.items())
        for name, prm in sig:
            if name in ("nef", "channels", "extra_channels") or name in skip:
                continue
            if name in kwargs:
                args[name] = kwargs[name]
            elif hasattr(self, name):
                args[name] = getattr(self, name)
            elif prm.default is not inspect.Parameter.empty:
                args[name] = prm.default
        return requested, extra, args

    def trace(self, nef, channels, extra_channels, rays, lod_idx=None, raymarch_type="voxel", num_steps=64, step_size=1.0,
              bg_color="white", stage="val", jitter=None):
        assert nef.grid is not None, "this tracer requires a grid"                      # :76
        N = rays.origins.shape[0]
        dev = rays.origins.device
        if lod_idx is None:
            lod_idx = nef.grid.num_lods - 1
        gkey = None
        if self.use_graphs and type(self).shade is PanopticPackedRFTracer.shade and raymarch_type in ("ray", "voxel"):
            from .graphs import GraphRunner
            if GraphRunner.eligible(self, nef, channels, extra_channels, rays, stage):
                if self._graphs is None:
                    self._graphs = GraphRunner()
                rb, gkey, jitter = self._graphs.run(self, nef, channels, rays, lod_idx, raymarch_type, num_steps, bg_color, stage, jitter)
                if rb is not None:
                    return rb           # else: no capacity known yet, or this batch overflowed it - the eager path below, same jitter
        marched = self._march(nef, rays, lod_idx, raymarch_type, num_steps, jitter)
        if gkey is not None:
            self._graphs.observe(gkey, marched[2].shape[0] * marched[6])
        return self._shade_marched(nef, channels, extra_channels, rays, marched, lod_idx, bg_color, stage)

    def _march(self, nef, rays, lod_idx, raymarch_type, num_steps, jitter=None, begun=None):
        """The ray march of trace() (:85-114) with everything that needs the host (the sample count sizes the packed tensors) -
        -> (ridx, pidx, samples, depths, deltas, boundary, k, ridx32, pack_start, ray_of_pack), the arguments of shade().
        begun: the state of grid.raymarch_voxel_begin() for these rays (render_packs queues the walk of a later pack early)."""
        dev = rays.origins.device
        kw = {"jitter": jitter} if jitter is not None else {}
        # voxel mode: grids of this package apply the travel filter of :88-108 inside the walk (pag_raymarch_voxel_*: same
        # strict `<` on the same fp32 difference) and hand back one pack per ray - no unique / repeat_interleave / mask passes
        filtered = raymarch_type == "voxel" and getattr(nef.grid, "accepts_max_travel", False)
        if filtered:
            kw["max_travel"] = self.ray_max_travel
        if begun is not None:
            ridx, pidx, samples, depths, deltas, boundary = nef.grid.raymarch_voxel_finish(begun, rays, num_steps)
        else:
            ridx, pidx, samples, depths, deltas, boundary = nef.grid.raymarch(                 # :85-86
                rays, level=nef.grid.active_lods[lod_idx], num_samples=num_steps, raymarch_type=raymarch_type, **kw)
        if samples.shape[0] and hasattr(nef, "prefetch_features") and (raymarch_type == "ray" or filtered) and self._prefetch:
            nef.prefetch_features(samples)        # first encode launch queued before the bookkeeping below (GPU idle otherwise)
        if raymarch_type == "voxel" and depths.numel() != 0 and not filtered:             # :88-108
            # drop nuggets further than ray_max_travel past the first hit of their ray (strict <)
            _, counts_per_ray = ridx.unique(return_counts=True)
            ray_start_idx = torch.cumsum(counts_per_ray, dim=0)
            ray_start_idx = torch.cat([torch.zeros(1, dtype=ray_start_idx.dtype, device=dev), ray_start_idx[:-1]])
            hit_depth = torch.take(depths[:, 0, 0], ray_start_idx)
            traveled_depth = depths[:, 0, 0] - torch.repeat_interleave(hit_depth, counts_per_ray)
            valid_mask = traveled_depth < self.ray_max_travel
            deltas = deltas.reshape(depths.shape)[valid_mask].reshape(-1, 1)
            boundary = boundary.reshape(depths.shape)[valid_mask].reshape(-1)
            ridx, pidx, samples, depths = ridx[valid_mask], pidx[valid_mask], samples[valid_mask], depths[valid_mask]
        k = samples.shape[1] if samples.dim() == 3 else 1                                   # samples per pack entry
        cache = getattr(nef.grid, "_pack_cache", None)
        if cache is not None and cache[0] is ridx:
            _, ridx32, pack_start, ray_of_pack = cache
        else:                                                                              # :114
            ridx32 = ridx.int() if k == 1 else ridx.int().repeat_interleave(k)            # one entry per SAMPLE
            pack_start, ray_of_pack = ops.packs_from_boundary(ridx32, boundary)
        return ridx, pidx, samples, depths, deltas, boundary, k, ridx32, pack_start, ray_of_pack

    def _shade_marched(self, nef, channels, extra_channels, rays, marched, lod_idx, bg_color, stage):
        ridx, pidx, samples, depths, deltas, _boundary, _k, ridx32, pack_start, ray_of_pack = marched
        outputs = self.shade(nef, channels, extra_channels, rays.dirs, rays.origins.shape[0], ridx, ridx32, pidx, samples, depths, deltas, pack_start,
                             ray_of_pack, lod_idx, bg_color, stage)
        return RenderBuffer(**outputs)

    _prefetch = True

    def render_packs(self, nef, packs, channels=None, extra_channels=None, **kwargs):
        """The reference's validation loop `for ray_pack in rays.split(render_batch): rb += pipeline(rays=ray_pack, ...)` (pc_nerf/trainer.py:637-649) with the
        VOXEL march of pack i + 1 running on a SECOND STREAM while pack i is shaded ('ray' marches take the plain loop, below): the march is a latency-bound walk (a 128^3 DDA per ray: ~80 us whether 8 000 or
        32 768 rays are in flight) whose sample count the host must read before it can size the packed tensors - in the plain loop the GPU idles through both, 116
        times per 720 x 1280 image at the reference's render_batch 8000.  Same launches on the same data in the same order per stream: the buffers are bit-identical
        to the plain loop's.  -> list of RenderBuffers, one per pack.  Inference only (torch.no_grad(); the caller's traces that need gradients go through
        forward())."""
        packs = list(packs)
        if not packs:
            return []
        own = type(self).shade is PanopticPackedRFTracer.shade and type(self).trace is PanopticPackedRFTracer.trace      # subclasses with their own trace / shade: the plain loop
        if torch.is_grad_enabled() or not packs[0].origins.is_cuda or not own:
            return [self.forward(nef, channels=channels, extra_channels=extra_channels, rays=p, **kwargs) for p in packs]
        requested, extra, args = self._resolve(nef, channels, extra_channels, kwargs, skip=("rays",))
        lod_idx = args.get("lod_idx")
        if lod_idx is None:
            lod_idx = nef.grid.num_lods - 1
        rm, steps, bg, stage, jitter = args["raymarch_type"], args["num_steps"], args["bg_color"], args.get("stage", "val"), args.get("jitter")
        if jitter is not None:
            return [self.forward(nef, channels=channels, extra_channels=extra_channels, rays=p, **kwargs) for p in packs]
        dev = packs[0].origins.device
        main = torch.cuda.current_stream(dev)
        side = getattr(self, "_march_stream", None)
        if side is None or side.device != dev:
            # high priority: its own hardware queue class (a process that has created many streams - graph captures, other side streams - otherwise
            # finds this one sharing a queue with the caller's stream, and the march serialises behind the shading it should run beside); the march
            # kernels are short walks that should never wait behind a 100 us decoder launch
            side = self._march_stream = torch.cuda.Stream(device=dev, priority=-1)
        out = []

        side.wait_stream(main)                           # the rays come from the caller's stream - waited for ONCE: a wait per pack would put the march of
                                                         # pack i + 1 behind the shading of pack i, which is queued on that stream by then

        # two deep for the voxel march of this package's grids: the walk of pack i + 2 is QUEUED (grid.raymarch_voxel_begin) before the host asks for the sample
        # count of pack i + 1, so the count is there when asked for; other marches (the 'ray' march: device-bound anyway) one deep
        split = rm == "voxel" and getattr(nef.grid, "accepts_max_travel", False) and hasattr(nef.grid, "raymarch_voxel_begin")
        if not split:
            # the 'ray' march is no latency-bound walk but a streaming writer (512 samples per ray: 0.55 GB per 32 768 rays): beside the shading it has nothing to
            # hide - the dense image is device-bound either way - and its stores push the level tables out of the L2 / MALL the gathers live on (measured:
            # 720 x 1280 dense image 241 -> 299 ms when the march overlapped, encode launches 10 % slower).  Plain loop.
            return [self.forward(nef, channels=channels, extra_channels=extra_channels, rays=p, **kwargs) for p in packs]

        def begin(pack):
            if not split:
                return None
            with torch.cuda.stream(side):
                return nef.grid.raymarch_voxel_begin(pack, steps, self.ray_max_travel)

        def march(pack, begun):
            with torch.cuda.stream(side):
                m = self._march(nef, pack, lod_idx, rm, steps, None, begun=begun)
                ev = torch.cuda.Event()
                ev.record(side)
            for t in m:
                if isinstance(t, torch.Tensor):
                    t.record_stream(main)                # allocated under the side stream, consumed (and released) on the caller's
            return m, ev
        self._prefetch = False                           # the first encode launch belongs to the shading stream
        try:
            b1 = begin(packs[1]) if len(packs) > 1 else None
            nxt = march(packs[0], begin(packs[0]))
            for i, pack in enumerate(packs):
                marched, ev = nxt
                main.wait_event(ev)
                out.append(self._shade_marched(nef, requested - extra, extra, pack, marched, lod_idx, bg, stage))     # queued on the caller's stream ...
                if i + 1 < len(packs):
                    b2 = begin(packs[i + 2]) if i + 2 < len(packs) else None
                    nxt = march(packs[i + 1], b1)        # ... and running while the host waits for the next pack's sample count
                    b1 = b2
        finally:
            self._prefetch = True
        return out

    def shade(self, nef, channels, extra_channels, ray_dirs, N, ridx, ridx32, pidx, samples, depths, deltas, pack_start, ray_of_pack,
              lod_idx, bg_color, stage):
        """Everything of trace() after the ray march (:117-205): the nef on the packed samples, compositing, the panoptic heads.
        -> dict channel -> tensor.  A function of tensors only (no host synchronisation, no shape that depends on device data), so
        that pagnerf_amd.graphs can capture it - forward and backward - into HIP graphs over static sample buffers."""
        dev = samples.device
        outputs = {}
        sample_channels = set(channels - self.render_channels)                             # :121-124
        sample_channels.update(["density"])
        # The panoptic channels use detached weights (:148-155), so they can be evaluated AFTER compositing, fused with
        # their per-ray weighted sum (nef.panoptic_composited): the [M, C] probabilities' gradient is never materialised.
        pan_req = [c for c in sorted(channels) if c in self.panoptic_channels]
        fuse_pan = bool(pan_req) and getattr(nef, "accepts_ray_index", False) and nef.can_fuse_panoptic(pan_req)
        if fuse_pan:
            sample_channels -= self.panoptic_channels
        if getattr(nef, "accepts_ray_index", False):      # per-ray view embedding gathered through ridx (no [M,3] dirs)
            feats = nef(coords=samples, ridx=ridx32, ray_dirs=ray_dirs, pidx=pidx, lod_idx=lod_idx, channels=sample_channels,
                        ray_packs=(pack_start, ray_of_pack))
        else:                                              # :117,:124
            feats = nef(coords=samples, ray_d=ray_dirs.index_select(0, ridx), pidx=pidx, lod_idx=lod_idx,
                        channels=sample_channels)
        sigma = feats["density"].reshape(-1)
        if self.ray_sparcity_reg > 0.0 and stage == "train":                               # :127-130
            per = torch.log(1.0 + 2 * sigma ** 2)
            ray_wise = torch.zeros(N, device=dev).scatter_add(0, ridx32.long(), per)
            outputs["ray_sparcity_loss"] = ray_wise.mean() * self.ray_sparcity_reg
        rgb = feats["rgb"].reshape(-1, 3) if "rgb" in channels else None
        dep = depths.reshape(-1) if "depth" in channels else None
        alpha, hit, out_rgb, out_depth, w = ops.composite(sigma, rgb, deltas.reshape(-1), dep, pack_start, ray_of_pack, N,
                                                          bg_white=(bg_color == "white"))  # :134-176
        outputs["alpha"] = alpha[:, None]
        outputs["hit"] = hit.view(torch.bool)          # the kernel writes 0 / 1 bytes: reinterpret, no cast launch
        if rgb is not None:
            outputs["rgb"] = out_rgb
        if dep is not None:
            outputs["depth"] = out_depth[:, None]
        alpha_d = alpha.detach()
        if fuse_pan:
            outputs.update(nef.panoptic_composited(samples, pan_req, w, alpha_d, ridx32, pack_start, ray_of_pack, N))
        else:
            for ch in pan_req:                                                              # :148-155,:178-182
                outputs[ch] = ops.composite_feats(feats[ch].reshape(-1, feats[ch].shape[-1]), w, alpha_d, pack_start, ray_of_pack, N)
        for ch in extra_channels:                                                          # :184-192
            # the reference composites extra channels with the LIVE alpha / transmittance (:192 passes the tensors of :135-138, not
            # the detached panoptic ones): their loss reaches the density through the weights as well as the channel itself
            f = nef(coords=samples, ray_d=ray_dirs.index_select(0, ridx), pidx=pidx, lod_idx=lod_idx, channels=ch)
            outputs[ch] = ops.composite_features(sigma, deltas.reshape(-1), f.reshape(-1, f.shape[-1]), ridx32, pack_start,
                                                 ray_of_pack, N)[0]
        for attr in ("_feat_cache", "_density_feats"):                                     # per-trace caches of the nef: nothing of this trace stays referenced
            if getattr(nef, attr, None) is not None:
                setattr(nef, attr, None)
        return outputs
