"""Ray sharding across the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" is
RCCL over xGMI on ROCm, "gloo" in the CPU tests).

The reference is single-GPU; this is the build's own addition.  Rays are
independent, so the hot path shards with no data-path exchange: every rank marches a contiguous block
of rays against replicated tables / decoders / occupancy.  Two collectives exist around it:
  * rendering: ONE all_gather of the per-rank [n_local, C] buffer (all channels packed side by side);
  * training:  the delta grid's table gradient (50.3 MB, complete early in the backward) is all-reduced asynchronously
    from a post-accumulate hook, the rest (50.3 MB main table + 0.14 MB decoders) as ONE flat all_reduce (GradSync).
Both are single large messages - on MI355X's point-to-point xGMI mesh a ring is bound by one link, so
fewer, larger collectives let RCCL spread traffic over all 7 links.
Options of GradSync: comm_dtype=torch.bfloat16 (bf16 messages, fp32 accumulation, two direct exchanges: _DirectReduce),
comm_dtype="auto" (fp32 or bf16 by regime), sparse=True / "exact" (only the union of the ranks' touched table rows travels:
SparseRows: the exact-sum exchange a quarter cheaper after the prune, no match for the bf16 one on xGMI).
"""
import time

import torch
import torch.distributed as dist

from .core import Rays, RenderBuffer


FORCE_COLLECTIVES = False      # tests: issue the collectives on a ONE-rank process group too (tests/test_gpu_rccl_single_rank.py: the only way to
                               # put RCCL itself under this module's calls on a one-GPU box)


def world_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _single():
    """True when there is nothing to exchange: one rank (unless a test forces the collectives onto the one-rank group)."""
    return world_info()[1] == 1 and not (FORCE_COLLECTIVES and dist.is_available() and dist.is_initialized())


_AVG_OK = {}        # backend -> bool: ReduceOp.AVG verified on this process group (first use)


def _avg_supported():
    """RCCL's ReduceOp.AVG, checked ONCE per backend with a tiny all-reduce of rank-valued numbers before any gradient depends on it:
    the averaging collective has no CPU (gloo) counterpart to test against, so the first use on a node verifies it and anything other
    than the exact mean - or an exception - sends every later call down the sum-then-divide path."""
    be = dist.get_backend()
    if be not in _AVG_OK:
        ok = False
        if be == "nccl":
            try:
                rank, world = dist.get_rank(), dist.get_world_size()
                t = torch.full((4,), float(rank + 1), device=torch.device("cuda", torch.cuda.current_device()))
                dist.all_reduce(t, op=dist.ReduceOp.AVG)
                ok = bool(torch.allclose(t.cpu(), torch.full((4,), (world + 1) / 2.0)))
            except Exception:
                ok = False
            flag = torch.tensor([1.0 if ok else 0.0], device=torch.device("cuda", torch.cuda.current_device()))
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)             # every rank takes the same branch
            ok = bool(flag.item() > 0.5)
        _AVG_OK[be] = ok
    return _AVG_OK[be]


def _reduce_op(average):
    """(op, divide afterwards?) - RCCL averages inside the collective (ReduceOp.AVG: no separate pass over a 50 MB table
    gradient) once _avg_supported() has verified it; gloo (the CPU tests) has no AVG: there the sum is divided by the world size."""
    if not average:
        return dist.ReduceOp.SUM, False
    if _avg_supported():
        return dist.ReduceOp.AVG, False
    return dist.ReduceOp.SUM, True


class _DirectReduce:
    """All-reduce of ONE large gradient as two direct exchanges in a reduced message dtype with fp32 accumulation:
         1. all_to_all_single: the gradient is cut in `world` chunks, rank r receives everyone's chunk r (bf16 on the wire),
         2. each rank sums its `world` received chunks in FP32 (and divides), rounds ONCE to the message dtype,
         3. all_gather_into_tensor: every rank receives every reduced chunk.
    Why this shape on MI355X: xGMI is a point-to-point mesh (7 links per GPU) - both phases send 1/world of the tensor to each peer
    over its own link at the same time, where a ring all-reduce is bound by one link; and the sum over ranks is formed in fp32, not in
    the message dtype hop by hop.  A 50.3 MB fp32 table gradient travels as 2 x 22 MB per rank at 8 ranks instead of 2 x 44 MB.
    Error per element: every rank's value rounded once to bf16 (2^-8 relative) + one rounding of the mean: |err| <= 2^-7 mean_r |g_r|
    (tests/test_shard_gloo.py checks the bound element by element)."""

    def __init__(self, grad, comm_dtype, average):
        _, world = world_info()
        self.grad, self.average, self.world = grad, average, world
        n = grad.numel()
        self.n, self.chunk = n, (n + world - 1) // world
        send = torch.zeros(world * self.chunk, device=grad.device, dtype=comm_dtype)
        send[:n] = grad.reshape(-1).to(comm_dtype)
        self.send = send
        self.recv = torch.empty_like(send)
        self.handle = dist.all_to_all_single(self.recv, self.send, async_op=True)

    def finish(self):
        self.handle.wait()
        red = self.recv.reshape(self.world, self.chunk).float().sum(0)          # fp32 accumulate
        if self.average:
            red /= self.world
        out = torch.empty(self.world * self.chunk, device=red.device, dtype=self.send.dtype)
        dist.all_gather_into_tensor(out, red.to(self.send.dtype))
        self.grad.copy_(out[:self.n].reshape(self.grad.shape))


def shard_bounds(n, rank, world):
    """Contiguous block partition of n rays: the first n % world ranks get one extra ray."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_rays(rays, rank=None, world=None):
    r, w = world_info()
    rank = r if rank is None else rank
    world = w if world is None else world
    lo, hi = shard_bounds(rays.origins.shape[0], rank, world)
    return rays[lo:hi]


def _as_labels(rb, label_channels):
    """rb with the channels named in label_channels replaced by their arg-max over the last dimension (int64 [rows])."""
    if not label_channels:
        return rb
    vals = {c: getattr(rb, c) for c in rb.channels if isinstance(getattr(rb, c), torch.Tensor)}
    for c in label_channels:
        if c in vals and vals[c].dim() > 1:
            vals[c] = vals[c].argmax(-1)
    return RenderBuffer(**vals)


def all_gather_render(rb, n_total, channels=None, label_channels=()):
    """Gather per-rank RenderBuffers (rows = this rank's ray block) into the full [n_total, .] buffers
    on every rank with one all_gather.  Bool channels travel as floats.
    label_channels: channels of which only the arg-max is needed (the semantic / instance images of validate(): trainer.py:717 and :740 take
    `argmax` of both) travel as ONE column instead of C - with the 6 + 200 probability columns of the panoptic heads a ray's message
    shrinks from 844 to 28 bytes.  They come back as int64 [n_total] labels; a single rank gets the same type."""
    rank, world = world_info()
    rb = _as_labels(rb, label_channels)
    if _single():
        return rb
    names = sorted(channels or [c for c in rb.channels if isinstance(getattr(rb, c), torch.Tensor) and getattr(rb, c).dim() > 0])
    cols = [getattr(rb, c).reshape(getattr(rb, c).shape[0], -1).float() for c in names]      # labels < 2^24: exact as floats
    widths = [c.shape[1] for c in cols]
    n_max = (n_total + world - 1) // world
    lo, hi = shard_bounds(n_total, rank, world)
    packed = torch.zeros(n_max, sum(widths), device=cols[0].device)
    packed[:hi - lo] = torch.cat(cols, 1)
    out = torch.empty(world * n_max, sum(widths), device=packed.device)
    dist.all_gather_into_tensor(out, packed)
    rows = torch.cat([out[r * n_max:r * n_max + (shard_bounds(n_total, r, world)[1] - shard_bounds(n_total, r, world)[0])]
                      for r in range(world)], 0)
    res, off = {}, 0
    for name, w in zip(names, widths):
        t = rows[:, off:off + w]
        src = getattr(rb, name)
        if src.dtype == torch.bool:
            t = t > 0.5
        elif not src.dtype.is_floating_point:
            t = t.round().to(src.dtype)
        res[name] = t.reshape(n_total, *src.shape[1:])
        off += w
    return RenderBuffer(**res)


SPARSE_DENSE_FILL = 0.5      # a level whose union of touched rows exceeds this share of its table travels whole
SPARSE_HEADROOM = 1.25       # bounded mode: slots per level = the recent maximum of the union count x this, in whole granules
SPARSE_GRANULE = 1024        # rows
SPARSE_HISTORY = 8           # steps whose counts size the slots
SPARSE_LAG = 2               # bounded mode: a step's slots come from the counts of the steps at least this many exchanges back (the same ones on every rank)


SPARSE_KERNELS = True      # GPU gradients: the mask / plan / pack / unpack passes of csrc/sparse.hip (False: the tensor-op form, as on CPU tensors)
_BIT_WEIGHTS = {}         # device -> uint8 [8] (a per-step torch.tensor(..., device=) would be a synchronising copy)


class SparseRows:
    """Touched-rows exchange of ONE table gradient [L, T, F] (levels x rows x features).

    After the first prune (configs/bup20/best.yaml:187 `prune_every: 201`, pc_nerf/trainer.py:362-366: three quarters of a run) a step's
    samples lie in the occupied cells only, and the coarse and middle lattice levels touch a small part of their 2^18 rows
    (scripts/touched_rows.py, profiles/r06_touched_rows_post_prune_8_ranks.json: levels 0 - 12 under 20 % even as the union over 8
    ranks; the fine levels are filled by hash collisions whatever the batch).  The reduce pass of the table gradient writes every row
    (pag_*_encode_bwd_set), so an untouched row is an exact zero on every rank and stays zero in the sum: only the union of the ranks'
    non-zero rows has to cross the links.  Per step and table:
      1. mask = any(grad != 0) per row, packed to bits (L x T / 8 bytes: 786 KB for 24 x 2^18), ONE all_gather, OR over the ranks:
         every rank holds the same union;
      2. per level: fill = |union| / T; levels above SPARSE_DENSE_FILL travel whole, the others as their union rows compacted in row
         order into a fixed number of slots;
      3. ONE collective over the concatenated slots of all levels (fp32 all-reduce, or the bf16 direct reduce of _DirectReduce);
      4. every row of the table gradient is rewritten from its slot (rows outside the union: zero).
    The values on the union rows are those of the dense all-reduce (same collective, same operands); rows outside it are exact zeros
    in both.

    How many slots a level gets must be known to the host before the collective is queued.  mode="exact": the host reads the per-level
    union counts (one device-to-host copy of L integers: the step's queue drains once) - nothing is ever dropped.  mode="bounded"
    (default): the slots follow the counts of EARLIER steps (recent maximum x SPARSE_HEADROOM over the steps SPARSE_LAG or more exchanges back: a FIXED
    lag, so that every rank sizes the same collective whatever the arrival time of its mailboxes - which have long landed in pinned memory by then), the
    host does not wait for the step in flight; union rows beyond a level's slots are set to zero on EVERY rank alike (the replicas stay identical),
    counted in `stats["dropped_rows"]` and reported by a warning, and the level's slots grow for the following steps.  reset() (call it when
    the regime changes: after nef.prune(), a new batch size or march type) sends the next steps whole until new counts are in."""

    def __init__(self, mode="bounded", dense_fill=SPARSE_DENSE_FILL, headroom=SPARSE_HEADROOM):
        assert mode in ("bounded", "exact")
        self.mode, self.dense_fill, self.headroom = mode, float(dense_fill), float(headroom)
        self.history = []            # per finished step: list of L union counts
        self._pending = []           # (event or None, pinned / cpu tensor [L + 1]) not yet read
        self._free_boxes = []        # pinned tensors read and ready for reuse
        self._caps = None            # per level: slots (T = whole level), host list
        self._caps_dev = self._offs_dev = None
        self.stats = dict(steps=0, exchanged_bytes=0, dense_bytes=0, bitmap_bytes=0, dropped_rows=0, last_fill=None, last_slots=None)
        self._warned = False

    def reset(self):
        self.history, self._pending, self._caps, self._caps_dev = [], [], None, None      # counts still in flight belong to the old regime

    # ---- host side: slots per level from the counts seen so far
    def _poll(self):
        """Take in the counts of the steps that lie SPARSE_LAG or more exchanges back - exactly those, on every rank: the slots of a step must be the same
        everywhere (they size the collective), so they cannot depend on WHEN a mailbox happens to arrive on this rank.  A report that old has normally landed;
        if the host has run that far ahead of its queue, it waits for it here (a run-ahead of SPARSE_LAG steps at most)."""
        while len(self._pending) >= SPARSE_LAG:
            ev, box = self._pending.pop(0)
            if ev is not None:
                ev.synchronize()
            vals = box.tolist()
            if ev is not None and len(self._free_boxes) < 8:
                self._free_boxes.append(box)
            self.history.append(vals[:-1])
            self.history = self.history[-SPARSE_HISTORY:]
            if vals[-1] > 0:
                self.stats["dropped_rows"] += int(vals[-1])
                if not self._warned:
                    import warnings
                    warnings.warn("pagnerf_amd.shard.SparseRows: %d union rows did not fit the slots sized from earlier steps and were zeroed on every rank "
                                  "(the slots grow within %d steps; call reset() when the regime changes, or use mode='exact')" % (int(vals[-1]), SPARSE_LAG + 1))
                    self._warned = True

    def _plan(self, T, L):
        if not self.history:
            return [T] * L
        caps = []
        for l in range(L):
            want = max(h[l] for h in self.history) * self.headroom
            c = int(-(-want // SPARSE_GRANULE) * SPARSE_GRANULE)
            caps.append(T if (want > self.dense_fill * T or c >= T) else max(c, SPARSE_GRANULE))
        return caps

    # ---- device side
    @staticmethod
    def union_mask(grad):
        """bool [L, T]: rows that are non-zero on ANY rank (bit-packed all_gather + OR)."""
        L, T, _ = grad.shape
        mask = (grad != 0).any(-1)
        pad = (-T) % 8
        if pad:
            mask = torch.cat([mask, mask.new_zeros(L, pad)], 1)
        w = _BIT_WEIGHTS.get(grad.device)
        if w is None:
            w = _BIT_WEIGHTS[grad.device] = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], device=grad.device, dtype=torch.uint8)
        bits = (mask.reshape(L, -1, 8).to(torch.uint8) * w).sum(-1, dtype=torch.uint8).reshape(-1)
        _, world = world_info()
        allb = torch.empty(world * bits.numel(), device=grad.device, dtype=torch.uint8)
        dist.all_gather_into_tensor(allb, bits)
        allb = allb.reshape(world, -1)
        u = allb[0].clone()
        for r in range(1, world):
            u |= allb[r]
        un = ((u[:, None] >> torch.arange(8, device=grad.device, dtype=torch.uint8)) & 1).bool().reshape(L, -1)[:, :T]
        return un, bits.numel()

    def start(self, grad, comm_dtype, average):
        """Queue the exchange of `grad` ([L, T, F], contiguous fp32); returns a token for finish().  GPU gradients take the four passes of csrc/sparse.hip
        (pag_sparse_rows_mask / _plan / _pack / _unpack: ~170 MB of traffic for a 50 MB table); CPU tensors (the gloo tests) and SPARSE_KERNELS = False the
        tensor-op form below - same slots, same values."""
        if grad.is_cuda and SPARSE_KERNELS:
            return self._start_kernels(grad, comm_dtype, average)
        L, T, F = grad.shape
        _, world = world_info()
        self._poll()
        union, bitmap_bytes = self.union_mask(grad)
        counts = union.sum(1)
        if self.mode == "exact":
            c_host = counts.tolist()                                   # the one host wait of this mode
            caps = [T if c > self.dense_fill * T else max(int(c), 1) for c in c_host]
        else:
            caps = self._plan(T, L)
        if caps != self._caps or self._caps_dev is None or self._caps_dev.device != grad.device:
            # two small host-to-device copies, only when the slots change (a per-step torch.tensor(..., device=) is a synchronising pageable copy)
            self._caps, self._caps_dev = caps, torch.tensor(caps, device=grad.device, dtype=torch.int32)
            self._offs_dev = torch.tensor([sum(caps[:l]) for l in range(L)], device=grad.device, dtype=torch.int64)
        cap_dev = self._caps_dev
        total = int(sum(caps))
        st = self.stats
        st["steps"] += 1
        st["dense_bytes"], st["bitmap_bytes"] = L * T * F * 4, bitmap_bytes
        half = comm_dtype is not None and comm_dtype != torch.float32
        if total == L * T:
            # every level whole (the first steps of the bounded mode, a dense regime): the plain exchange of the gradient itself; the counts still
            # go to the mailbox so that the following steps can size their slots
            self._mail(torch.cat([counts, counts.new_zeros(1)]), grad)
            if half:
                handle, divide = _DirectReduce(grad, comm_dtype, average), False
            else:
                op, divide = _reduce_op(average)
                handle = dist.all_reduce(grad, op=op, async_op=True)
            st["exchanged_bytes"], st["last_slots"], st["whole_levels"] = L * T * F * (2 if half else 4), total, L
            return (grad, None, None, handle, divide, world)
        offs = self._offs_dev
        whole = (cap_dev >= T)[:, None]
        member = union | whole                                         # a whole level: every row has a slot, its own
        pos = torch.cumsum(member, 1, dtype=torch.int32)
        valid = member & (pos <= cap_dev[:, None])
        slot = torch.where(valid, offs[:, None] + (pos - 1).long(), torch.full((), total, device=grad.device, dtype=torch.int64)).reshape(-1)
        row_of_slot = torch.zeros(total + 1, device=grad.device, dtype=torch.int64)
        row_of_slot.scatter_(0, slot, torch.arange(L * T, device=grad.device, dtype=torch.int64))
        filled = torch.zeros(total + 1, device=grad.device, dtype=torch.bool)
        filled[slot] = True
        flat = grad.reshape(L * T, F)
        buf = flat.index_select(0, row_of_slot[:total]) * filled[:total, None].to(flat.dtype)
        dropped = (union & ~valid).sum()
        self._mail(torch.cat([counts, dropped[None]]), grad)
        if half:
            handle, divide = _DirectReduce(buf, comm_dtype, average), False
        else:
            op, divide = _reduce_op(average)
            handle = dist.all_reduce(buf, op=op, async_op=True)
        st["exchanged_bytes"], st["last_slots"], st["whole_levels"] = total * F * (2 if half else 4), total, int(sum(1 for c in caps if c >= T))
        return (grad, buf, slot, handle, divide, world)

    def _start_kernels(self, grad, comm_dtype, average):
        from . import ops
        L, T, F = grad.shape
        W = (T + 31) // 32
        dev = grad.device
        _, world = world_info()
        self._poll()
        st_ = ops.L.stream
        bits = torch.empty(L * W, device=dev, dtype=torch.int32)
        ops._call("pag_sparse_rows_mask", grad.data_ptr(), L, T, F, bits.data_ptr(), st_())
        allb = torch.empty(world * L * W, device=dev, dtype=torch.int32)
        dist.all_gather_into_tensor(allb, bits)
        allb = allb.reshape(world, -1)
        union = allb[0].clone()
        for r in range(1, world):
            union |= allb[r]
        prefix = torch.empty(L * W, device=dev, dtype=torch.int32)
        counts = torch.empty(L + 1, device=dev, dtype=torch.int64)

        def plan(caps):
            if caps != self._caps or self._caps_dev is None or self._caps_dev.device != dev:
                self._caps, self._caps_dev = caps, torch.tensor(caps, device=dev, dtype=torch.int32)
                self._offs_dev = torch.tensor([sum(caps[:l]) for l in range(L)], device=dev, dtype=torch.int64)
            ops._call("pag_sparse_rows_plan", union.data_ptr(), L, T, self._caps_dev.data_ptr(), prefix.data_ptr(), counts.data_ptr(), st_())
        if self.mode == "exact":
            plan([0] * L)                                              # counts only
            c_host = counts[:L].tolist()                               # the one host wait of this mode
            caps = [T if c > self.dense_fill * T else max(int(c), 1) for c in c_host]
        else:
            caps = self._plan(T, L)
        plan(caps)
        total = int(sum(caps))
        st = self.stats
        st["steps"] += 1
        st["dense_bytes"], st["bitmap_bytes"] = L * T * F * 4, L * W * 4
        half = comm_dtype is not None and comm_dtype != torch.float32
        self._mail(counts, grad)
        if total == L * T:                                             # every level whole: the plain exchange of the gradient itself
            if half:
                handle, divide = _DirectReduce(grad, comm_dtype, average), False
            else:
                op, divide = _reduce_op(average)
                handle = dist.all_reduce(grad, op=op, async_op=True)
            st["exchanged_bytes"], st["last_slots"], st["whole_levels"] = L * T * F * (2 if half else 4), total, L
            return (grad, None, None, handle, divide, world)
        buf = torch.zeros(total, F, device=dev)
        ops._call("pag_sparse_rows_pack", grad.data_ptr(), L, T, F, union.data_ptr(), prefix.data_ptr(), self._caps_dev.data_ptr(), self._offs_dev.data_ptr(), buf.data_ptr(), st_())
        if half:
            handle, divide = _DirectReduce(buf, comm_dtype, average), False
        else:
            op, divide = _reduce_op(average)
            handle = dist.all_reduce(buf, op=op, async_op=True)
        st["exchanged_bytes"], st["last_slots"], st["whole_levels"] = total * F * (2 if half else 4), total, int(sum(1 for c in caps if c >= T))
        return (grad, buf, ("kernels", union, prefix, self._caps_dev, self._offs_dev), handle, divide, world)

    def _mail(self, box_dev, grad):
        """Per-level union counts (+ rows dropped) of this step towards the host, without waiting: pinned copy + event on a device, a plain copy on CPU."""
        box_dev = box_dev.to(torch.int64)
        if grad.is_cuda:
            # pinned words come from a small pool (pin_memory() is a host allocation of its own: ~1 ms per call)
            box = self._free_boxes.pop() if self._free_boxes else torch.empty(box_dev.numel(), dtype=torch.int64).pin_memory()
            if box.numel() != box_dev.numel():
                box = torch.empty(box_dev.numel(), dtype=torch.int64).pin_memory()
            box.copy_(box_dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        else:
            box, ev = box_dev.clone(), None
        self._pending.append((ev, box))

    @staticmethod
    def finish(token):
        grad, buf, slot, handle, divide, world = token
        if isinstance(handle, _DirectReduce):
            handle.finish()
        else:
            handle.wait()
            if divide:
                (grad if buf is None else buf).div_(world)
        if buf is None:
            return
        if isinstance(slot, tuple):                                    # the kernel form: every row rewritten from its slot in one pass
            from . import ops
            _, union, prefix, caps_dev, offs_dev = slot
            L, T, F = grad.shape
            ops._call("pag_sparse_rows_unpack", buf.data_ptr(), L, T, F, union.data_ptr(), prefix.data_ptr(), caps_dev.data_ptr(), offs_dev.data_ptr(), grad.data_ptr(), ops.L.stream())
            return
        ext = torch.cat([buf, buf.new_zeros(1, buf.shape[1])], 0)
        grad.copy_(ext.index_select(0, slot).reshape(grad.shape))


def allreduce_grads(params, average=True, big=1 << 20, comm_dtype=None):
    """Gradients of >= `big` elements (the tables, ~50 MB each) are all-reduced in place, one message each; everything
    smaller (decoders, poses: ~0.14 MB) travels as ONE flat all_reduce.  No staging copy of the large tensors.
    comm_dtype (e.g. torch.bfloat16): the large gradients go through _DirectReduce (reduced-precision messages, fp32 accumulation)."""
    rank, world = world_info()
    if _single():
        return
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    large = [g for g in grads if g.numel() >= big and g.is_contiguous()]
    small = [g for g in grads if not (g.numel() >= big and g.is_contiguous())]
    op, divide = _reduce_op(average)
    if comm_dtype is not None and comm_dtype != torch.float32:
        direct = [_DirectReduce(g, comm_dtype, average) for g in large]
        handles = []
    else:
        direct = []
        handles = [dist.all_reduce(g, op=op, async_op=True) for g in large]
    if small:
        flat = torch.cat([g.reshape(-1).float() for g in small])
        dist.all_reduce(flat, op=op)
        if divide:
            flat /= world
        off = 0
        for g in small:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
    for d in direct:
        d.finish()
    for g, h in zip(large, handles):
        h.wait()
        if divide:
            g /= world


AUTO_BUS_GBS = 250.0       # bus bandwidth of a large-message RCCL all-reduce assumed by comm_dtype="auto" (200 - 300 GB/s on an 8-GPU xGMI node)
AUTO_FACTOR = 4.0          # "auto" switches to bf16 direct reduce when the step is shorter than AUTO_FACTOR x the predicted exposed fp32 exchange
AUTO_WARM = 3              # steps observed before "auto" decides (the first ones hold captures / allocator warm-up)


def predicted_exchange_ms(nbytes, world, bus_gbs=AUTO_BUS_GBS):
    """Time of a ring all-reduce of `nbytes` per rank at `bus_gbs` of bus bandwidth: 2 (W-1)/W x bytes / bandwidth (the model of DESIGN section 6)."""
    return 2.0 * (world - 1) / max(world, 1) * nbytes / (bus_gbs * 1e9) * 1e3


class GradSync:
    """Gradient exchange of a training step with the early part overlapped with the rest of the backward.

    `early` parameters (the delta grid's table: its gradient is complete after the panoptic heads' backward, before the
    colour / density decoders and the main grid run theirs) are all-reduced asynchronously from a post-accumulate hook;
    finish() waits for them, exchanges everything else as one flat all-reduce and averages.  On the xGMI mesh the 50 MB
    early message therefore travels while the GPU still computes; only the main table (produced last) is exposed."""

    def __init__(self, params, early=(), average=True, comm_dtype=None, big=1 << 20, bus_gbs=AUTO_BUS_GBS, sparse=False):
        """comm_dtype=torch.bfloat16: table-sized gradients travel as bf16 messages with fp32 accumulation (_DirectReduce): half the
        bytes of the fp32 all-reduce and direct per-link transfers; the default (None) keeps RCCL's fp32 all-reduce.
        comm_dtype="auto": chosen by REGIME after AUTO_WARM steps - the exposed part of the fp32 exchange is the largest table-sized gradient
        that is not `early` (the main table: ~0.35 ms at 8 ranks whatever the batch); when the measured step (maximum over the ranks, agreed
        through one all-reduce so that every rank switches at the same step) is shorter than AUTO_FACTOR x that prediction - the post-prune
        regimes, where a 0.5 - 1 ms step would spend a third or more of its time waiting for the table - the bf16 direct reduce takes over;
        long steps (the dense regime) keep the exact fp32 all-reduce.  `auto_decision` records what was decided and from which numbers."""
        self.params = list(params)
        self.average = average
        self._auto = comm_dtype == "auto"
        self.bus_gbs = float(bus_gbs)
        self.auto_decision = None
        self._marks, self._intervals = [], []
        if self._auto:
            comm_dtype = None
        self.comm_dtype = comm_dtype if comm_dtype not in (None, torch.float32) else None
        self.big = big
        # sparse: False | True / "bounded" | "exact" - table-shaped gradients ([L, T, F] with >= `big` elements) travel as the union of the ranks'
        # touched rows (SparseRows); one state per parameter, `sparse_stats()` reports what the last step moved
        self.sparse_mode = {False: None, None: None, True: "bounded", "bounded": "bounded", "exact": "exact"}[sparse]
        self._sparse = {}
        self.early = [p for p in early]
        self._early_ids = {id(p) for p in self.early}
        self._handles = []
        self._hooks = []
        if not _single():
            for p in self.early:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._launch))

    def _sparse_state(self, p):
        st = self._sparse.get(id(p))
        if st is None:
            st = self._sparse[id(p)] = SparseRows(self.sparse_mode)
        return st

    def _is_table(self, g):
        return self.sparse_mode is not None and g.dim() == 3 and g.numel() >= self.big and g.is_contiguous() and g.dtype == torch.float32

    def reset_sparse(self):
        """Call when the set of touched rows changes regime (after nef.prune(), another batch size or march type): the next steps travel whole."""
        for st in self._sparse.values():
            st.reset()

    def sparse_stats(self):
        """Per table-shaped parameter (in `params` order): what its last exchange moved."""
        return [dict(self._sparse[id(p)].stats) for p in self.params if id(p) in self._sparse]

    def _launch(self, p):
        if p.grad is None:
            return
        if self._is_table(p.grad):
            self._handles.append((p, ("sparse", self._sparse_state(p).start(p.grad, self.comm_dtype, self.average)), None))
            return
        if self.comm_dtype is not None and p.grad.numel() >= self.big and p.grad.is_contiguous():
            self._handles.append((p, _DirectReduce(p.grad, self.comm_dtype, self.average), None))      # phase 1 in flight under the backward
        else:
            op, divide = _reduce_op(self.average)
            self._handles.append((p, dist.all_reduce(p.grad, op=op, async_op=True), divide))

    def _decide(self, world):
        """comm_dtype="auto": one collective decision from the step cadence seen so far (called from finish(), after this step's exchange)."""
        dev = next((p.grad.device for p in self.params if p.grad is not None), torch.device("cpu"))
        # the cadence is DEVICE time where there is a device (an event per finish(), read once at the decision): a host that runs ahead of its
        # queue would otherwise report its own issue rate and take a dense run for a short one
        if dev.type == "cuda":
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self._marks.append(ev)
        else:
            self._marks.append(time.perf_counter())
        if len(self._marks) < AUTO_WARM + 2:
            return
        if dev.type == "cuda":
            self._marks[-1].synchronize()
            iv = [a.elapsed_time(b) for a, b in zip(self._marks[1:-1], self._marks[2:])]      # the first interval holds one-time set-up
        else:
            iv = [(b - a) * 1e3 for a, b in zip(self._marks[1:-1], self._marks[2:])]
        self._intervals = iv
        step_ms = sum(iv) / len(iv)
        grads = [p.grad for p in self.params if p.grad is not None and id(p) not in self._early_ids and p.grad.numel() >= self.big]
        exposed = max([g.numel() * g.element_size() for g in grads], default=0)
        # BOTH numbers are agreed over the ranks (maximum): a rank whose set of gradient-carrying parameters differed would otherwise pick
        # another collective than its peers and hang
        t = torch.tensor([step_ms, float(exposed)], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        step_ms, exposed = float(t[0].item()), int(t[1].item())
        pred = predicted_exchange_ms(exposed, world, self.bus_gbs)
        use_bf16 = exposed > 0 and step_ms < AUTO_FACTOR * pred
        self.comm_dtype = torch.bfloat16 if use_bf16 else None
        self.auto_decision = dict(step_ms=round(step_ms, 4), exposed_bytes=int(exposed), predicted_fp32_exchange_ms=round(pred, 4), factor=AUTO_FACTOR,
                                  bus_gbs=self.bus_gbs, comm_dtype="bf16" if use_bf16 else "fp32", decided_after_steps=len(self._marks))
        self._auto = False

    def finish(self):
        _, world = world_info()
        if _single():
            return
        rest = [p for p in self.params if id(p) not in self._early_ids or not any(q is p for q, _, _ in self._handles)]
        if self.sparse_mode is not None:
            tables = [p for p in rest if p.grad is not None and self._is_table(p.grad)]
            for p in tables:
                self._handles.append((p, ("sparse", self._sparse_state(p).start(p.grad, self.comm_dtype, self.average)), None))
            rest = [p for p in rest if not any(q is p for q in tables)]
        allreduce_grads(rest, average=self.average, big=self.big, comm_dtype=self.comm_dtype)
        for p, h, divide in self._handles:
            if isinstance(h, tuple) and h[0] == "sparse":
                SparseRows.finish(h[1])
                continue
            if isinstance(h, _DirectReduce):
                h.finish()
                continue
            h.wait()
            if divide:
                p.grad /= world
        self._handles = []
        if self._auto:
            self._decide(world)

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


def render_sharded(pipeline, rays, channels, label_channels=(), **kwargs):
    """Every rank renders its block of `rays` and receives the full image buffers (label_channels: see all_gather_render)."""
    n = rays.origins.shape[0]
    local = shard_rays(rays)
    rb = pipeline(rays=local, channels=channels, **kwargs)
    return all_gather_render(rb, n, channels=None, label_channels=label_channels)
