"""HIP-graph execution of the train step's post-march part behind the tracer API (PanopticPackedRFTracer(use_graphs=True)).

Why: with a few hundred thousand samples per step (the voxel march after the first prune - three quarters of a BUP20 training run,
best.yaml:34,165 - or the 256-ray plumbing configuration) the kernels of a step are 10 - 100 us each and the step is bound by the
host: ~35 launches of 15 - 25 us of Python each, plus the GPU idling while the host waits for the sample count that sizes every
tensor of the step (the reference formulation - boolean-mask indexing in wisp's OctreeAS.raymarch - has the same read-back).

How: the march kernels write into upper-bound buffers without the host learning the count (ops.march_into); pag_pad_packed appends
filler samples that belong to no ray's pack up to a fixed CAPACITY chosen from the counts of the previous steps, so every tensor after
the march has a shape that does not depend on device data (the per-ray kernels never see the fillers; the per-sample gradient tensors
of the compositing backward are zero past the real samples, so the fillers carry no gradient); tracer.shade() - the nef on the packed samples,
compositing, the panoptic heads - is captured ONCE per capacity (_Graphed: forward graph + backward graph(s); our C-ABI launches go
to torch's current stream, which is the capture stream) and replayed from then on: one graph launch forward, one backward when the
caller's `loss.backward()` reaches it.  The count is read from the pinned mailbox AFTER the forward graph has been queued; only if it
exceeds the capacity (rare: the capacity follows the recent maximum with 2 % head-room, in geometric buckets with hysteresis) is the
result discarded and the eager path run with the same jitter - the replay itself ran truncated at the capacity (clamped pack table).

Forward values are bit-identical to the eager path (the fillers take part in no per-ray sum); gradients agree up to the fp32
summation order of the per-wave weight-gradient slabs (the tile -> wave assignment depends on the padded sample count).

Ownership: the outputs handed out are copies; p.grad may alias the capture's static gradients until the next trace of the
configuration, which first turns such a p.grad into a private copy (_GraphedFn).  With more than one rank
the backward is captured as two graphs so that gradient hooks fire between them (_Graphed, `split`).

Pose optimisation (rays that require a gradient: pc_nerf/ba_pipeline.py:85-92, every step of a configs/bup20/best.yaml run -
optimize_extrinsics with extrinsics_epoch_end 900 > epochs 800): the per-ray transform stays outside the capture, in ordinary
autograd; the captured function takes the ray origins / directions as two more static inputs that require a gradient (the samples are
attached to them by ops.ray_samples inside the capture), and the backward graph hands d origins / d dirs [N,3] back as two more static
outputs - the position gradient of the main grid's encoder, the per-ray sums of pag_ray_sample_grad and the view embedding's gradient
all replay with the rest of the backward.

Not taken (the tracer falls back to eager): torch.no_grad() / stage != 'train', extra channels, ray_sparcity_reg > 0, foreign grids.
"""
import collections
import os
import weakref

import torch
import torch.nn as nn

from . import ops
from .core import RenderBuffer

GRANULE = 8192          # capacities are multiples of GRANULE * k samples
HEADROOM = 1.02         # capacity >= the recent maximum count * HEADROOM
HISTORY = 8             # steps whose counts decide whether the capacity must GROW
CAP_STEPS = 32          # capacities are multiples of 1/CAP_STEPS of the power of two below them: <= 3.1 % of filler samples, and a batch
                        # size that drifts by a per cent or two (ordinary batch-to-batch noise, the slow drift after a prune) stays in one bucket
SHRINK_WINDOW = 64      # the capacity only shrinks when every count of this many steps would fit the smaller one (hysteresis)
MAX_BUCKETS = 2         # captured capacities kept per configuration (each owns a private memory pool: every activation of `cap` samples)


def _round_capacity(want, k):
    """The smallest multiple of (2^floor(log2 want) / CAP_STEPS, itself rounded to whole granules) that holds `want` samples."""
    gran = GRANULE * k
    step = max(gran, (1 << (max(int(want), 1).bit_length() - 1)) // CAP_STEPS // gran * gran)
    return (int(want) + step - 1) // step * step


class _PostMarch(nn.Module):
    """tracer.shade() as a function of the static sample tensors (the integer index tensors are static attributes of `buf`)."""

    def __init__(self, nef, tracer, buf, capacity, channels, lod_idx, bg_color, stage):
        super().__init__()
        self.nef = nef
        self._t = (tracer, buf)                       # plain attributes: not sub-modules
        self.capacity, self.channels, self.lod_idx, self.bg_color, self.stage = capacity, frozenset(channels), lod_idx, bg_color, stage
        self.names = None

    def forward(self, samples, depths, deltas, ray_dirs, origins=None):
        """origins (pose optimisation): [N,3] requiring a gradient - only its SHAPE and its place in the autograd graph matter (the
        samples' values are the march kernel's); with it the samples carry d / d origins and d / d dirs as in the eager path
        (grids.OccupancyBLAS.raymarch: ops.ray_samples)."""
        tracer, buf = self._t
        cap, k, N = self.capacity, buf.k, buf.N
        ne = cap // k
        smp = samples.reshape(ne, k, 3)
        dep = depths.reshape(ne, k)
        if origins is not None:
            smp = ops.ray_samples(origins, ray_dirs, smp, dep, buf.pack_start_c, ops._ray_iota(N, samples.device), ridx=buf.ridx_sample[:cap])
        # pack_start_c = min(pack_start, capacity): these launches are queued before the host has seen the sample count, and a batch that
        # overflows the capacity (its result is discarded afterwards) must not send a per-ray kernel past the capacity-sized tensors
        out = tracer.shade(self.nef, set(self.channels), set(), ray_dirs, N, buf.ridx_entry[:ne], buf.ridx_sample[:cap], buf.pidx[:ne], smp, dep,
                           deltas, buf.pack_start_c, ops._ray_iota(N, samples.device), self.lod_idx, self.bg_color, self.stage)
        if self.names is None:
            self.names = sorted(out)
        return tuple(out[n] for n in self.names)


def _fresh(statics):
    """Copies of the capture's static tensors in ordinary (caching-allocator) memory: ONE launch."""
    outs = [torch.empty_like(o, memory_format=torch.contiguous_format) for o in statics]
    if outs:
        ops.copy_batch(outs, list(statics))       # one launch (six runtime copies of 4 KB - 3 MB cost 4.7 us each)
    return outs


class _GraphedFn(torch.autograd.Function):
    """One autograd node per backward GROUP of a captured configuration (one group = the whole backward by default; two when the
    capture is split, see _Graphed).  _Graphed.__call__ replays the forward graph, then creates the nodes.

    Ownership (every output freshly allocated per call; trainer.py:426 `zero_grad(set_to_none=True)` is the reference's own
    usage but not a contract): the outputs handed to the caller are COPIES of the static outputs (a RenderBuffer of step k is untouched
    by step k + 1).  The gradients returned to autograd are the static buffers themselves - AccumulateGrad adopts such a tensor as
    `p.grad` when the parameter has none, which saves a 100 MB copy per step - and therefore, before the NEXT trace replays the forward
    graph (whose temporaries may share pool blocks with them), any `p.grad` that still shares their storage (gradient accumulation
    over several traces, `zero_grad(set_to_none=False)`, a GradScaler that unscaled in place) is first detached from the capture: it
    becomes a private copy (_Graphed.detach_param_grads).  A caller that follows the reference's loop (`set_to_none=True`, one trace per
    step) never pays for that copy.  What is NOT protected: a gradient tensor the caller took out of `p.grad` and kept beyond the next
    trace of the same configuration without cloning it."""

    @staticmethod
    def forward(ctx, runner, gi, *params):
        """params: the group's real parameters, then the live tensors of the capture's gradient-carrying arguments the group reaches
        (pose optimisation: ray directions / origins)."""
        ctx.runner, ctx.gi = runner, gi
        ctx.set_materialize_grads(False)      # outputs the loss does not use arrive as None in backward(), not as zero tensors filled per step
        grp = runner.groups[gi]
        ctx.generation = runner.generation
        idx = grp.out_idx + (runner.nondiff_idx if gi == 0 else [])
        outs = _fresh([runner.outs[i] for i in idx])
        ctx.mark_non_differentiable(*outs[len(grp.out_idx):])
        return tuple(outs)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *grads):
        r = ctx.runner
        grp = r.groups[ctx.gi]
        if ctx.generation != r.generation:
            raise RuntimeError("pagnerf_amd.graphs: the forward graph of this configuration was replayed again before this backward ran - the "
                               "activations it saved are gone.  Call backward() on a trace before tracing the same configuration again "
                               "(or construct the tracer with use_graphs=False / 'static').")
        r.detach_param_grads(grp)         # (already done before the forward replay; a p.grad assigned since then is caught here)
        # upstream gradients -> the backward graph's static inputs: ONE multi-tensor copy (they were five ~3 us launches in a 1.4 ms step);
        # an output the loss does not use arrives as None: its static gradient is zeroed once and stays zero (the graph only reads it)
        dst, src, zero = [], [], []
        for i, (static, g) in enumerate(zip(grp.gouts, grads)):
            if g is None:
                if i not in grp.zeroed:
                    zero.append(static)
                    grp.zeroed.add(i)
            else:
                dst.append(static)
                src.append(g if (g.dtype == static.dtype and g.shape == static.shape) else g.to(static.dtype).reshape(static.shape))
                grp.zeroed.discard(i)
        if zero:
            torch._foreach_zero_(zero)
        if dst:
            ops.copy_batch(dst, src)
        grp.bwd.replay()
        # parameters: the static gradients themselves (see above); argument gradients ([N,3] d dirs / d origins of a pose-optimisation
        # step): copies - their consumer is the caller's own autograd graph, which may keep or accumulate them
        n_p = len(grp.params)
        return (None, None) + tuple(g.detach() for g in grp.gins[:n_p]) + tuple(_fresh(grp.gins[n_p:]))


class _Group:
    """One backward graph: the differentiable outputs `out_idx` (indices into _Graphed.outs) -> gradients of `params`."""
    __slots__ = ("out_idx", "gouts", "bwd", "gins", "params", "zeroed", "arg_sel")


class _Graphed:
    """Forward and backward HIP graphs of one _PostMarch module over its static arguments.

    torch.cuda.make_graphed_callables is not used: it differentiates with respect to the module's real parameters, and when an EARLIER
    eager step's autograd graph is still referenced by the caller (its RenderBuffer / loss usually are, until the next assignment) the
    parameters' AccumulateGrad nodes of that step - created on the default stream - are reused by the engine, which then synchronises
    the capture stream with the default stream in the middle of the backward capture: hipStreamEndCapture crashed (ROCm 7.2).  Here
    the captured function runs on ALIASES of the parameters (fresh leaves sharing their storage: torch.func.functional_call), so the
    captured backward never touches a pre-existing autograd node; the real parameters are inputs of _GraphedFn and receive the static
    gradients through ordinary autograd outside any capture.

    split (names of outputs, e.g. the panoptic channels): the backward is captured as TWO graphs - the gradients that flow from the
    `split` outputs, then those of the remaining outputs - and each graph sits behind its own autograd node, the split one created
    last so that the engine runs it first (each with a memory pool of its own: correct in either order).  Whatever a caller hangs on the parameters between the two (shard.GradSync's post-accumulate
    hook: the delta grid's table gradient is complete after the panoptic heads' backward and starts its all-reduce while the colour /
    density decoders and the main grid are still running) fires as in an eager backward.  Same kernels, same values."""

    def __init__(self, mod, args, split=(), grad_args=()):
        """grad_args: indices into `args` of static tensors the captured function is differentiated with respect to as well (pose
        optimisation: the ray directions and the origin stand-in); __call__ then takes the live tensors that stand behind them."""
        self.mod = mod
        named = [(n, p) for n, p in mod.named_parameters() if p.requires_grad]
        params = [p for _, p in named]
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            self.alias = {n: p.detach().requires_grad_(True) for n, p in named}
            # fresh leaves on the static argument storage (what the march / padding launches write before every replay)
            args = tuple(a.detach().requires_grad_(True) if i in grad_args else a for i, a in enumerate(args))
        self.args = args
        self.grad_args = tuple(grad_args)
        leaves = list(self.alias.values()) + [args[i] for i in grad_args]

        def run():
            return torch.func.functional_call(mod, self.alias, args)

        def groups_of(outs):
            """Output indices per backward graph, in the order the graphs are captured AND replayed: the split outputs' graph first."""
            diff = [i for i, o in enumerate(outs) if o.requires_grad]
            late = [i for i in diff if mod.names[i] in split]
            rest = [i for i in diff if mod.names[i] not in split]
            return [g for g in (late, rest) if g] or [[]]
        with torch.cuda.stream(side):
            for _ in range(2):          # warm-up: lazy initialisations (cached index tensors, workspaces) happen outside the capture
                outs = run()
                gs = groups_of(outs)
                for j, idx in enumerate(gs):
                    if idx:
                        torch.autograd.grad([outs[i] for i in idx], leaves, [torch.zeros_like(outs[i]) for i in idx], allow_unused=True,
                                            retain_graph=j + 1 < len(gs))
                del outs
        cur.wait_stream(side)
        torch.cuda.synchronize()
        pool = torch.cuda.graph_pool_handle()
        self.fwd = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.fwd, pool=pool):
            self.outs = run()
        self.nondiff_idx = [i for i, o in enumerate(self.outs) if not o.requires_grad]
        self.groups = []
        gs = groups_of(self.outs)
        for j, idx in enumerate(gs):
            grp = _Group()
            grp.out_idx = idx
            grp.gouts = [torch.zeros_like(self.outs[i]) for i in idx]
            grp.zeroed = set()             # indices of gouts known to hold zeros (set by _GraphedFn.backward)
            grp.bwd = torch.cuda.CUDAGraph()
            if idx:
                # ONE backward graph shares the forward's pool (its temporaries reuse the activations autograd releases on the way: replayed
                # strictly after the forward, as captured).  TWO backward graphs get a pool each: a pool hands memory freed by an earlier
                # capture to a later one, which is only sound when replays follow the capture order - and which of the two autograd nodes
                # runs first (or at all: a loss without the panoptic terms) is the engine's / the caller's choice, not ours.
                with torch.cuda.graph(grp.bwd, pool=pool if len(gs) == 1 else torch.cuda.graph_pool_handle()):
                    gins = torch.autograd.grad([self.outs[i] for i in idx], leaves, grp.gouts, allow_unused=True, retain_graph=j + 1 < len(gs))
            else:
                gins = [None] * len(leaves)
            used = [i for i, g in enumerate(gins) if g is not None]
            grp.params = [params[i] for i in used if i < len(params)]     # a parameter this group's outputs do not depend on is not an input of its node
            grp.arg_sel = [i - len(params) for i in used if i >= len(params)]      # positions in grad_args of the arguments this group reaches
            grp.gins = [gins[i] for i in used]
            self.groups.append(grp)
        self.generation = 0
        self.unaliased = 0                 # p.grad tensors that had to be detached from the capture's static gradients (diagnostics / tests)
        torch.cuda.synchronize()

    def detach_param_grads(self, grp=None):
        """A p.grad that still shares storage with the capture's static gradients becomes a private copy.  Called BEFORE the forward
        replay: the static gradients live in the capture's memory pool, where the forward graph's temporaries may occupy the same
        blocks - a replay of the FORWARD already scribbles over them, not just the next backward."""
        for g_ in ([grp] if grp is not None else self.groups):
            for p, g in zip(g_.params, g_.gins):
                pg = p.grad
                if pg is not None and pg.untyped_storage().data_ptr() == g.untyped_storage().data_ptr():
                    p.grad = pg.clone()
                    self.unaliased += 1

    def __call__(self, live_args=()):
        """live_args: one live tensor per grad_args entry (same order) - the nodes hang the captured gradients on them."""
        assert len(live_args) == len(self.grad_args)
        self.detach_param_grads()
        self.fwd.replay()
        self.generation += 1
        res = [None] * len(self.outs)
        # autograd runs the node created LAST first: the groups are listed in the order their backward graphs should run
        for gi in reversed(range(len(self.groups))):
            grp = self.groups[gi]
            outs = _GraphedFn.apply(self, gi, *grp.params, *[live_args[j] for j in grp.arg_sel])
            for i, o in zip(grp.out_idx + (self.nondiff_idx if gi == 0 else []), outs):
                res[i] = o
        return tuple(res)


class _State:
    def __init__(self):
        self.counts = collections.deque(maxlen=HISTORY)
        self.long = collections.deque(maxlen=SHRINK_WINDOW)
        self.cap = None
        self.buf = None
        self.dirs = None
        self.orig0 = None           # pose optimisation: [N,3] stand-in for the ray origins inside the capture (shape + autograd position only)
        self.buckets = collections.OrderedDict()

    def see(self, count):
        self.counts.append(int(count))
        self.long.append(int(count))

    def capacity(self, buf):
        """Capacity for the next step: grows at once when the recent counts (+ head-room) no longer fit, shrinks only after SHRINK_WINDOW
        steps all of which would fit a smaller bucket; geometric buckets (_round_capacity) - every NEW capacity costs a capture (two
        warm-up steps, two synchronisations) and a private memory pool, so ordinary batch-to-batch noise must not move it."""
        want = _round_capacity(int(max(self.counts) * HEADROOM) + 1, buf.k)
        if self.cap is None or want > self.cap:
            self.cap = want
            self.long.clear()
        elif len(self.long) == self.long.maxlen:
            small = _round_capacity(int(max(self.long) * HEADROOM) + 1, buf.k)
            if small < self.cap:
                self.cap = small
                self.long.clear()
        self.cap = min(buf.cap, self.cap)
        return self.cap

    def bucket(self, key, make=None):
        """Least-recently-used cache of at most MAX_BUCKETS captured capacities (the evicted graphs release their memory pool)."""
        b = self.buckets.get(key)
        if b is not None:
            self.buckets.move_to_end(key)
            return b
        if make is None:
            return None
        while len(self.buckets) >= MAX_BUCKETS:
            self.buckets.popitem(last=False)
        b = self.buckets[key] = make()
        return b


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
    def _split(tracer):
        """Names of the outputs whose backward is captured as a graph of its own (see _Graphed): the panoptic channels when more than one
        rank trains (shard.GradSync's early all-reduce), or when PAG_GRAPH_SPLIT=1 / tracer.graph_split forces it (tests)."""
        force = getattr(tracer, "graph_split", None)
        if force is None:
            force = os.environ.get("PAG_GRAPH_SPLIT", "") not in ("", "0")
        multi = torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1
        return tuple(sorted(tracer.panoptic_channels)) if (force or multi) else ()

    @staticmethod
    def eligible(tracer, nef, channels, extra_channels, rays, stage):
        if not torch.is_grad_enabled() or stage != "train" or extra_channels or tracer.ray_sparcity_reg > 0.0:
            return False
        if not rays.origins.is_cuda:
            return False
        if not getattr(nef, "accepts_ray_index", False) or not getattr(nef.grid, "accepts_max_travel", False):
            return False
        return True

    PARAM_RESCAN = 64       # steps between full walks of the module tree (a step in between looks the known parameter slots up directly)
    MAX_SLOT_CACHES = 4     # nefs whose parameter slots are remembered at once (two nefs alternating on one tracer do not thrash)

    @staticmethod
    def _structure(nef, mods):
        """Cheap structural fingerprint of the nef and its remembered sub-modules: a sub-module or parameter that was added, removed or
        swapped in changes the number of entries of some module's `_parameters` dict or the identity of a child (one pass over ~25 small
        dicts, no recursion) - the slots are then walked again at once instead of at the next periodic rescan."""
        out = [(len(m._parameters), *m._modules.values()) for m in mods]          # modules compare by identity
        out.append((len(nef._parameters), *nef._modules.values()))
        return out

    def _param_sig(self, nef):
        """((data_ptr, requires_grad), ...) of the nef's parameters - part of the key: a capture bakes the storages in.  `nef.parameters()`
        walks the whole module tree (~90 us of Python per step, a sixth of a post-prune rgb-only step): the (module, name) slots are
        remembered per nef and read directly; a replaced Parameter object shows up at once (its slot is read every step), a module or
        parameter that was added / removed through the structural fingerprint, anything else at the next full walk.  The cache holds the
        nef itself only weakly (its sub-modules strongly: they die with the entry, which goes when the nef is gone or the cache is full)."""
        caches = self.__dict__.setdefault("_slot_caches", collections.OrderedDict())
        for k in [k for k, c in caches.items() if c[0]() is None]:
            del caches[k]
        cache = caches.get(id(nef))
        if cache is not None and (cache[0]() is not nef or cache[2] <= 0 or self._structure(nef, cache[3]) != cache[4]):
            cache = None                                                      # weak reference: a new nef at a recycled address is a different nef
        if cache is None:
            mods = [m for m in nef.modules() if m is not nef]
            slots = [(m, n) for m in mods for n, p in m._parameters.items() if p is not None]
            own = [n for n, p in nef._parameters.items() if p is not None]
            cache = caches[id(nef)] = [weakref.ref(nef), slots, self.PARAM_RESCAN, mods, self._structure(nef, mods), own]
            while len(caches) > self.MAX_SLOT_CACHES:
                caches.popitem(last=False)
        caches.move_to_end(id(nef))
        cache[2] -= 1
        try:
            sig = [(m._parameters[n].data_ptr(), m._parameters[n].requires_grad) for m, n in cache[1]]
            sig += [(nef._parameters[n].data_ptr(), nef._parameters[n].requires_grad) for n in cache[5]]
            return tuple(sig)
        except (KeyError, AttributeError):          # a slot vanished: walk again
            caches.pop(id(nef), None)
            return self._param_sig(nef)

    def _key(self, tracer, nef, channels, rays, lod_idx, raymarch_type, num_steps, bg_color, stage):
        lw = nef.lod_weights
        grids = [nef.grid] + ([nef.delta_grid] if hasattr(nef, "delta_grid") else [])
        return (id(nef), raymarch_type, int(rays.origins.shape[0]), int(num_steps), frozenset(channels), lod_idx, bg_color, stage, nef.precision,
                nef.training, id(lw), lw._version, tuple((id(g.tables), g.tables.dtype, g.rounds_coords(), g.blas_level) for g in grids),
                float(rays.dist_min), float(rays.dist_max), float(tracer.ray_max_travel), str(rays.origins.device),
                self._param_sig(nef), self._split(tracer), bool(rays.origins.requires_grad or rays.dirs.requires_grad))

    def observe(self, key, count):
        """An eager step of this configuration saw `count` samples: the first capacities are chosen from it."""
        self._state(key).see(count)

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
        pose = bool(rays.origins.requires_grad or rays.dirs.requires_grad)
        if pose and st.orig0 is None:
            st.orig0 = torch.zeros(N, 3, device=dev)
        buf = st.buf
        cap = st.capacity(buf)
        bits = None if g._all_occupied else g.blas_bits
        if bits is not None and bits.device != dev:
            g.blas_bits = bits = bits.to(dev)
        coarse = g._coarse_bits(bits) if (bits is not None and raymarch_type == "voxel") else None
        # count -> offsets + padding to `cap` + the directions into the static tensor (one launch) -> pack
        mailbox, jitter = ops.march_into(buf, rays.origins, rays.dirs, rays.dist_min, rays.dist_max, num_steps, jitter=jitter,
                                         occupancy_bits=bits, blas_level=g.blas_level,
                                         max_travel=tracer.ray_max_travel if raymarch_type == "voxel" else None, occupancy_coarse_bits=coarse,
                                         pad_capacity=cap, dirs_out=st.dirs)
        args = (buf.samples[:cap], buf.depths[:cap], buf.deltas[:cap], st.dirs) + ((st.orig0,) if pose else ())
        live = (rays.dirs, rays.origins) if pose else ()
        if tracer.use_graphs == "static":
            # same static, padded buffers and optimistic count check - but the post-march part runs as ordinary eager launches: what the graph
            # path gains by never waiting for the sample count (the host runs ahead of the device) without a capture, for callers whose
            # backward must stay an ordinary autograd pass.  A batch that overflows `cap` runs TRUNCATED (pack_start_c, see _PostMarch) and is
            # discarded below: the launches stay inside the capacity-sized tensors.
            mod = st.bucket(("static", cap), lambda: _PostMarch(nef, tracer, buf, cap, channels, lod_idx, bg_color, stage))
            ops.SAMPLES_HINT, ops.TAIL_ZERO = max(st.counts), True
            try:
                outs = mod(*args[:3], *live) if pose else mod(*args)       # eager launches: the live rays take the place of the static copies
            finally:
                ops.SAMPLES_HINT, ops.TAIL_ZERO = None, False
            self.replays += 1
            M = self._count(mailbox, buf)
            st.see(M)
            if M > cap:
                self.overflows += 1
                return None, key, jitter
            return RenderBuffer(**dict(zip(mod.names, outs))), key, jitter
        graphed = st.bucket(cap)
        if graphed is None:
            # the count must be known to be <= cap before the capture's warm-up runs shade() on these buffers for real
            M = self._count(mailbox, buf)
            mailbox = None
            st.see(M)
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
            while len(st.buckets) >= MAX_BUCKETS:       # release the evicted capture's pool BEFORE the new one allocates its own
                st.buckets.popitem(last=False)
            gc.collect()
            torch.cuda.synchronize()
            mod = _PostMarch(nef, tracer, buf, cap, channels, lod_idx, bg_color, stage)
            mod.train(nef.training)
            ops.SAMPLES_HINT = M            # launch heuristics see the real count, not the padded capacity (same split as the eager path)
            ops.TAIL_ZERO = True            # per-sample tensors that the per-pack kernels fill start as zeros: the fillers carry no gradient
            try:
                graphed = st.bucket(cap, lambda: _Graphed(mod, args, split=self._split(tracer), grad_args=(3, 4) if pose else ()))
            finally:
                ops.SAMPLES_HINT, ops.TAIL_ZERO = None, False
            self.captures += 1
        outs = graphed(live)
        self.replays += 1
        if mailbox is not None:
            M = self._count(mailbox, buf)
            st.see(M)
            if M > cap:                     # the batch did not fit: the replay ran on a batch truncated at `cap` samples - discard it
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
