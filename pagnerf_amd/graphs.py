"""HIP-graph execution of the train step's post-march part behind the tracer API (PanopticPackedRFTracer(use_graphs=True)).

Why: with a few hundred thousand samples per step (the voxel march after the first prune - three quarters of a BUP20 training run,
best.yaml:34,165 - or the 256-ray plumbing configuration) the kernels of a step are 10 - 100 us each and the step is bound by the
host: ~35 launches of 15 - 25 us of Python each, plus the GPU idling while the host waits for the sample count that sizes every
tensor of the step (the reference formulation - boolean-mask indexing in wisp's OctreeAS.raymarch - has the same read-back).

How: the march kernels write into upper-bound buffers without the host learning the count (ops.march_into); pag_pad_packed appends
filler samples that belong to no ray's pack up to a fixed CAPACITY chosen from the counts of the previous steps, so every tensor after
the march has a shape that does not depend on device data (the per-ray kernels never see the fillers; the per-sample gradient tensors
of the compositing backward are zero past the real samples, so the fillers carry no gradient); tracer.shade() - the nef on the packed samples,
compositing, the panoptic heads - is captured ONCE per capacity with torch.cuda.make_graphed_callables (forward graph + backward
graph; our C-ABI launches go to torch's current stream, which is the capture stream) and replayed from then on: one graph launch
forward, one backward when the caller's `loss.backward()` reaches it.  The count is read from the pinned mailbox AFTER the forward
graph has been queued; only if it exceeds the capacity (rare: the capacity follows the recent maximum with 2 % head-room) is the
result discarded and the eager path run with the same jitter.

Forward values are bit-identical to the eager path (the fillers take part in no per-ray sum); gradients agree up to the fp32
summation order of the per-wave weight-gradient slabs (the tile -> wave assignment depends on the padded sample count).

Not taken (the tracer falls back to eager): torch.no_grad() / stage != 'train', extra channels, ray_sparcity_reg > 0, rays that
require a gradient (pose optimisation), foreign grids, N > 1 ranks (the early all-reduce of the delta table needs its gradient before
the backward graph ends).
"""
import collections

import torch
import torch.nn as nn

from . import ops
from .core import RenderBuffer

GRANULE = 8192          # capacities are multiples of GRANULE * k samples
HEADROOM = 1.02         # capacity >= the recent maximum count * HEADROOM
HISTORY = 8             # steps whose counts decide the capacity


class _PostMarch(nn.Module):
    """tracer.shade() as a function of the static sample tensors (the integer index tensors are static attributes of `buf`)."""

    def __init__(self, nef, tracer, buf, capacity, channels, lod_idx, bg_color, stage):
        super().__init__()
        self.nef = nef
        self._t = (tracer, buf)                       # plain attributes: not sub-modules
        self.capacity, self.channels, self.lod_idx, self.bg_color, self.stage = capacity, frozenset(channels), lod_idx, bg_color, stage
        self.names = None

    def forward(self, samples, depths, deltas, ray_dirs):
        tracer, buf = self._t
        cap, k, N = self.capacity, buf.k, buf.N
        ne = cap // k
        smp = samples.reshape(ne, k, 3)
        dep = depths.reshape(ne, k)
        out = tracer.shade(self.nef, set(self.channels), set(), ray_dirs, N, buf.ridx_entry[:ne], buf.ridx_sample[:cap], buf.pidx[:ne], smp, dep,
                           deltas, buf.pack_start, ops._ray_iota(N, samples.device), self.lod_idx, self.bg_color, self.stage)
        if self.names is None:
            self.names = sorted(out)
        return tuple(out[n] for n in self.names)


class _GraphedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, runner, *params):
        ctx.runner = runner
        ctx.set_materialize_grads(False)      # outputs the loss does not use arrive as None in backward(), not as zero tensors filled per step
        runner.fwd.replay()
        outs = tuple(o.detach() for o in runner.outs)
        ctx.mark_non_differentiable(*[o for o, s in zip(outs, runner.outs) if not s.requires_grad])
        return outs

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *grads):
        r = ctx.runner
        # upstream gradients -> the backward graph's static inputs: ONE multi-tensor copy (they were five ~3 us launches in a 1.4 ms step);
        # an output the loss does not use arrives as None: its static gradient is zeroed once and stays zero (the graph only reads it)
        dst, src, zero = [], [], []
        for i, (static, g) in enumerate(zip(r.gouts, grads)):
            if static is None:
                continue
            if g is None:
                if i not in r.zeroed:
                    zero.append(static)
                    r.zeroed.add(i)
            elif static.data_ptr() != g.data_ptr():
                dst.append(static)
                src.append(g if (g.dtype == static.dtype and g.shape == static.shape) else g.to(static.dtype).reshape(static.shape))
                r.zeroed.discard(i)
        if zero:
            torch._foreach_zero_(zero)
        if dst:
            torch._foreach_copy_(dst, src)
        r.bwd.replay()
        return (None,) + tuple(g.detach() if g is not None else None for g in r.gins)


class _Graphed:
    """Forward and backward HIP graphs of one _PostMarch module over its static arguments.

    torch.cuda.make_graphed_callables is not used: it differentiates with respect to the module's real parameters, and when an EARLIER
    eager step's autograd graph is still referenced by the caller (its RenderBuffer / loss usually are, until the next assignment) the
    parameters' AccumulateGrad nodes of that step - created on the default stream - are reused by the engine, which then synchronises
    the capture stream with the default stream in the middle of the backward capture: hipStreamEndCapture crashed (ROCm 7.2).  Here
    the captured function runs on ALIASES of the parameters (fresh leaves sharing their storage: torch.func.functional_call), so the
    captured backward never touches a pre-existing autograd node; the real parameters are inputs of _GraphedFn and receive the static
    gradients through ordinary autograd outside any capture."""

    def __init__(self, mod, args):
        self.mod, self.args = mod, args
        named = [(n, p) for n, p in mod.named_parameters() if p.requires_grad]
        self.params = [p for _, p in named]
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            self.alias = {n: p.detach().requires_grad_(True) for n, p in named}
        leaves = list(self.alias.values())

        def run():
            outs = torch.func.functional_call(mod, self.alias, args)
            return outs, [o for o in outs if o.requires_grad]
        with torch.cuda.stream(side):
            for _ in range(2):          # warm-up: lazy initialisations (cached index tensors, workspaces) happen outside the capture
                outs, req = run()
                torch.autograd.grad(req, leaves, [torch.zeros_like(o) for o in req], allow_unused=True)
                del outs, req
        cur.wait_stream(side)
        torch.cuda.synchronize()
        pool = torch.cuda.graph_pool_handle()
        self.fwd, self.bwd = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.fwd, pool=pool):
            self.outs, req = run()
        self.gouts = [torch.zeros_like(o) if o.requires_grad else None for o in self.outs]
        self.zeroed = set()             # indices of gouts known to hold zeros (set by _GraphedFn.backward)
        with torch.cuda.graph(self.bwd, pool=pool):
            self.gins = torch.autograd.grad(req, leaves, [g for g in self.gouts if g is not None], allow_unused=True)
        torch.cuda.synchronize()

    def __call__(self):
        return _GraphedFn.apply(self, *self.params)


class _State:
    def __init__(self):
        self.counts = collections.deque(maxlen=HISTORY)
        self.buf = None
        self.dirs = None
        self.buckets = {}


class GraphRunner:
    """Per-tracer cache of march buffers and captured graphs, keyed by everything that is baked into a capture."""

    MAX_STATES = 4      # configurations (key below) kept alive at once: each owns march buffers and graphs worth GBs at full size

    def __init__(self):
        self.states = collections.OrderedDict()
        self.replays = self.captures = self.overflows = 0

    def _state(self, key):
        """The state of `key`, most recently used last; the least recently used configuration is dropped (buffers, graphs and their
        memory pool) when a new one would exceed MAX_STATES - e.g. after the tables were replaced or the optimiser re-created the leaves."""
        st = self.states.get(key)
        if st is None:
            while len(self.states) >= self.MAX_STATES:
                self.states.popitem(last=False)
            st = self.states[key] = _State()
        else:
            self.states.move_to_end(key)
        return st

    @staticmethod
    def eligible(tracer, nef, channels, extra_channels, rays, stage):
        if not torch.is_grad_enabled() or stage != "train" or extra_channels or tracer.ray_sparcity_reg > 0.0:
            return False
        if rays.origins.requires_grad or rays.dirs.requires_grad or not rays.origins.is_cuda:
            return False
        if not getattr(nef, "accepts_ray_index", False) or not getattr(nef.grid, "accepts_max_travel", False):
            return False
        if tracer.use_graphs != "static" and torch.distributed.is_available() and torch.distributed.is_initialized() \
                and torch.distributed.get_world_size() > 1:
            return False            # captured backward: the gradient hooks of shard.GradSync would all fire after the whole graph (no overlap)
        return True

    def _key(self, tracer, nef, channels, rays, lod_idx, raymarch_type, num_steps, bg_color, stage):
        lw = nef.lod_weights
        grids = [nef.grid] + ([nef.delta_grid] if hasattr(nef, "delta_grid") else [])
        return (id(nef), raymarch_type, int(rays.origins.shape[0]), int(num_steps), frozenset(channels), lod_idx, bg_color, stage, nef.precision,
                nef.training, id(lw), lw._version, tuple((id(g.tables), g.tables.dtype, g.rounds_coords(), g.blas_level) for g in grids),
                float(rays.dist_min), float(rays.dist_max), float(tracer.ray_max_travel), str(rays.origins.device),
                tuple(p.data_ptr() for p in nef.parameters()))

    def observe(self, key, count):
        """An eager step of this configuration saw `count` samples: the first capacities are chosen from it."""
        self._state(key).counts.append(int(count))

    def run(self, tracer, nef, channels, rays, lod_idx, raymarch_type, num_steps, bg_color, stage, jitter):
        """-> (RenderBuffer | None, key, jitter used).  None: take the eager path (and call observe(key, M) afterwards)."""
        key = self._key(tracer, nef, channels, rays, lod_idx, raymarch_type, num_steps, bg_color, stage)
        st = self._state(key)
        if not st.counts:
            return None, key, jitter
        g = nef.grid
        dev = rays.origins.device
        N = rays.origins.shape[0]
        if st.buf is None:
            if raymarch_type == "ray":
                st.buf = ops.MarchBuffers("ray", N, num_steps, 1, dev)
            else:
                st.buf = ops.MarchBuffers("voxel", N, int(ops.L.load().pag_raymarch_voxel_nugget_capacity(g.blas_level)), int(num_steps), dev)
            st.dirs = torch.empty(N, 3, device=dev)
        buf = st.buf
        gran = GRANULE * buf.k
        want = int(max(st.counts) * HEADROOM) + 1
        cap = min(buf.cap, max(gran, (want + gran - 1) // gran * gran))
        bits = None if g._all_occupied else g.blas_bits
        if bits is not None and bits.device != dev:
            g.blas_bits = bits = bits.to(dev)
        coarse = g._coarse_bits(bits) if (bits is not None and raymarch_type == "voxel") else None
        mailbox, jitter = ops.march_into(buf, rays.origins, rays.dirs, rays.dist_min, rays.dist_max, num_steps, jitter=jitter,
                                         occupancy_bits=bits, blas_level=g.blas_level,
                                         max_travel=tracer.ray_max_travel if raymarch_type == "voxel" else None, occupancy_coarse_bits=coarse)
        buf.pad_to(cap)
        if st.dirs.data_ptr() != rays.dirs.data_ptr():
            st.dirs.copy_(rays.dirs)
        args = (buf.samples[:cap], buf.depths[:cap], buf.deltas[:cap], st.dirs)
        if tracer.use_graphs == "static":
            # same static, padded buffers and optimistic count check - but the post-march part runs as ordinary eager launches: what the graph
            # path gains by never waiting for the sample count (the host runs ahead of the device) without a capture, for callers whose
            # backward must stay an ordinary autograd pass (gradient hooks: the early all-reduce of shard.GradSync at N > 1)
            mod = st.buckets.get(("static", cap))
            if mod is None:
                mod = st.buckets[("static", cap)] = _PostMarch(nef, tracer, buf, cap, channels, lod_idx, bg_color, stage)
            ops.SAMPLES_HINT, ops.TAIL_ZERO = max(st.counts), True
            try:
                outs = mod(*args)
            finally:
                ops.SAMPLES_HINT, ops.TAIL_ZERO = None, False
            self.replays += 1
            M = self._count(mailbox, buf)
            st.counts.append(M)
            if M > cap:
                self.overflows += 1
                return None, key, jitter
            return RenderBuffer(**dict(zip(mod.names, outs))), key, jitter
        graphed = st.buckets.get(cap)
        if graphed is None:
            # the count must be known to be <= cap before the capture's warm-up runs shade() on these buffers for real
            M = self._count(mailbox, buf)
            mailbox = None
            st.counts.append(M)
            if M > cap:
                return None, key, jitter
            # tensors an earlier EAGER step left cached on the nef / grids (feature cache, density features, pack tables) would be
            # released in the middle of the capture when the captured forward overwrites them: drop them first, outside any capture
            import gc
            for attr in ("_density_feats", "_feat_cache", "_prefetched"):
                if hasattr(nef, attr):
                    setattr(nef, attr, None)
            for grid in [nef.grid] + ([nef.delta_grid] if hasattr(nef, "delta_grid") else []):
                grid._pack_cache = None
            gc.collect()
            torch.cuda.synchronize()
            mod = _PostMarch(nef, tracer, buf, cap, channels, lod_idx, bg_color, stage)
            mod.train(nef.training)
            ops.SAMPLES_HINT = M            # launch heuristics see the real count, not the padded capacity (same split as the eager path)
            ops.TAIL_ZERO = True            # per-sample tensors that the per-pack kernels fill start as zeros: the fillers carry no gradient
            try:
                graphed = _Graphed(mod, args)
            finally:
                ops.SAMPLES_HINT, ops.TAIL_ZERO = None, False
            st.buckets[cap] = graphed
            self.captures += 1
        outs = graphed()
        self.replays += 1
        if mailbox is not None:
            M = self._count(mailbox, buf)
            st.counts.append(M)
            if M > cap:                     # the batch did not fit: the replay ran on a truncated batch - discard it
                self.overflows += 1
                return None, key, jitter
        return RenderBuffer(**dict(zip(graphed.mod.names, outs))), key, jitter

    @staticmethod
    def _count(mailbox, buf):
        M = ops._poll_count(mailbox) if mailbox is not None else -1
        if mailbox is not None and M >= 0:
            ops._release_mailbox(mailbox)
        if M < 0:                            # no mailbox (polling disabled) or timed out: synchronising read-back of the TRUE count
            M = int(buf.counts.sum().item())
        return M
