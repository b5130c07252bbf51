"""Linear-assignment instance losses on the rendered `inst_embedding` (the consumer directly after the
hot path), with the cost matrix built on the GPU.

  LinAssignmentThingsLoss  <- loss/lin_assignment_things.py::LinAssignmentThingsLoss
  LinAssignmentLoss        <- loss/lin_assignment.py::LinAssignmentLoss

The reference builds cost[l, :] = -(sum of the probabilities of the rays labelled l) / (count + 1e-4)
with one masked sum + one device-to-host copy per label (lin_assignment_things.py:31-33).  Here ONE
launch (pag_label_sums) produces all per-label sums and counts where the probabilities already are;
only the [K, I] matrix goes to the host, where SciPy's Hungarian solver runs exactly as in the
reference (:45), and the relabelling (:47-53) is a table lookup on the device.  The optional ID-range
cost (utils/outlier_rejection.py:8-51) needs per-id 3-D centres (:56-71): the same kernel on the
[P,3] points.
"""
import os

import numpy as np
import scipy.optimize
import torch
import torch.nn as nn

from . import _lib as L
from . import ops
from .ops import render_loss, NllTerm          # noqa: F401  (the trainer's per-ray objective as two launches)


def label_sums(values, labels_gt, label_list, col0=0, row_mask=None):
    """-> (sums f32 [K, C-col0], counts i32 [K]) of `values` [P,C] rows grouped by labels_gt == label_list[k]."""
    ops._check_gpu(values, labels_gt)
    dev = values.device
    if values.dtype not in (torch.float32, torch.bfloat16):
        values = values.float()
    values = values.detach()
    if values.stride(-1) != 1:
        values = values.contiguous()
    P, C = values.shape
    labels_gt = labels_gt.detach().contiguous().long()
    lab = torch.as_tensor(label_list, dtype=torch.int64, device=dev).contiguous()
    K = lab.shape[0]
    sums = torch.zeros(K, C - col0, device=dev)
    counts = torch.zeros(K, device=dev, dtype=torch.int32)
    mask = row_mask.detach().contiguous().to(torch.uint8) if row_mask is not None else None
    if K and P:
        ops._call("pag_label_sums", values.data_ptr(), L.dtype_code(values), P, values.stride(0), col0, C - col0, L.ptr(labels_gt),
                  L.ptr(mask), L.ptr(lab), K, L.ptr(sums), L.ptr(counts), L.stream())
    return sums, counts


def cost_matrix(prob, labels_gt, labels, col0=0):
    """float64 [K, C-col0] numpy cost matrix: -(per-label fp32 sum / (count + 1e-4)), as :31-33 builds it."""
    sums, counts = label_sums(prob, labels_gt, labels, col0=col0)
    cost = -(sums / (counts.long() + 1e-4)[:, None])             # int64 + python float -> fp32, as in the reference
    return cost.cpu().numpy().astype(np.float64)


def _id_range_allowed(centers_x, num_ids, frame_min_length=0.3, max_num_inst_at_x=30, id_margin=30):
    """bool [K, num_ids] on centers_x's device: the ids inside [lo(x), lo(x) + margin] of each instance's x position (utils/outlier_rejection.py:8-51)."""
    slope = (max_num_inst_at_x + id_margin) / frame_min_length
    x_limit = (num_ids - id_margin) / slope
    x = (-centers_x + 1) / 2
    lo = torch.clamp(slope * (x % x_limit), 0, num_ids - 1).long()
    hi = torch.clamp(lo + id_margin, 0, num_ids - 1)
    ids = torch.arange(num_ids, device=centers_x.device)[None, :]
    return (lo[:, None] <= ids) & (ids <= hi[:, None])


def id_range_cost(cost, centers_x, frame_min_length=0.3, max_num_inst_at_x=30, id_margin=30):
    """utils/outlier_rejection.py:8-51: ids outside [lo(x), lo(x)+margin] of an instance's x position cost 10000.
    cost float64 [K,num_ids] (modified in place), centers_x f32 tensor [K]."""
    allowed = _id_range_allowed(centers_x, cost.shape[1], frame_min_length, max_num_inst_at_x, id_margin)
    cost[~allowed.cpu().numpy()] = 10000
    return cost


def _lookup(labels_gt, labels, targets, default):
    """per-ray relabelling: labels[i] -> targets[i], every other value -> default (a device table lookup)."""
    lo, hi = int(min(labels)), int(max(labels))
    lut = torch.full((hi - lo + 1,), default, dtype=labels_gt.dtype)
    lut[torch.tensor([l - lo for l in labels], dtype=torch.long)] = torch.as_tensor(targets, dtype=labels_gt.dtype)
    lut = lut.to(labels_gt.device)
    inside = (labels_gt >= lo) & (labels_gt <= hi)
    return torch.where(inside, lut[(labels_gt - lo).clamp(0, hi - lo)], torch.full_like(labels_gt, default))


class _AssignNLL(torch.autograd.Function):
    """loss [B, P] of LinAssignmentThingsLoss.forward (:56-82) for the assignment (labels i64 [B, R] sorted, targets i64 [B, R], info i32 [B, 2]: the first
    info[b, 0] labels of image b map to their targets, every other positive id to 1): per image ONE forward launch pair (virtual labels, arg-max, `any wrong`,
    -log) and one backward launch instead of ~25 tensor ops."""

    @staticmethod
    def forward(ctx, prob, labels_gt, stuff_mask, labels, targets, info):
        B, P, I = prob.shape
        dev = prob.device
        virt = torch.empty(B, P, device=dev, dtype=torch.int64)
        valid = torch.empty(B, P, device=dev, dtype=torch.uint8)
        loss = torch.empty(B, P, device=dev)
        wrong = torch.zeros(B, device=dev, dtype=torch.int32)
        ops._call("pag_assign_nll_fwd", prob.data_ptr(), B, P, prob.stride(0), prob.stride(1), I, labels_gt.data_ptr(),
                  stuff_mask.data_ptr() if stuff_mask is not None else None, labels.data_ptr(), targets.data_ptr(), info.data_ptr(), labels.shape[1], 1,
                  virt.data_ptr(), loss.data_ptr(), valid.data_ptr(), wrong.data_ptr(), L.stream())
        ctx.save_for_backward(prob, virt, valid, wrong)
        ctx.mark_non_differentiable(virt)
        return loss, virt

    @staticmethod
    def backward(ctx, g, _g_virt=None):
        prob, virt, valid, wrong = ctx.saved_tensors
        B, P, I = prob.shape
        g = g.contiguous().float()
        d = torch.empty(B, P, I, device=prob.device)
        ops._call("pag_assign_nll_bwd", prob.data_ptr(), B, P, prob.stride(0), prob.stride(1), I, virt.data_ptr(), valid.data_ptr(), wrong.data_ptr(), g.data_ptr(),
                  d.data_ptr(), L.stream())
        return d, None, None, None, None, None


class LinAssignmentThingsLoss(nn.Module):
    def __init__(self, outlier_rejection=False, min_distance=0.2, max_distance=0.5, *args, **kwargs):
        super().__init__()
        self.outlier_rejection = outlier_rejection
        self.min_distance, self.max_distance = min_distance, max_distance
        self._ws = None          # device scratch + pinned host mirrors of the one-synchronisation path, keyed by (B, P, I, device)
        self.fast_path = True
        # solver: "device" (default; pag_assign_solve - SciPy's algorithm on the GPU, the step never waits for the host) or "scipy" (one copy + wait, SciPy per
        # image on the host).  Same assigned columns either way (tests/test_gpu_loss.py).  PAG_ASSIGN_SOLVER overrides the default.
        self.solver = kwargs.pop("solver", None) or os.environ.get("PAG_ASSIGN_SOLVER", "device")
        self.last_virtual_labels = None

    @torch.no_grad()
    def create_virtual_gt_with_linear_assignment(self, inst_probabilities, labels_gt, points_3d=None):
        """[P,I] probabilities, [P] gt ids (0 = stuff) -> [P] virtual labels (:23-54).  Rows with gt <= 0 get 0."""
        things = labels_gt > 0
        n_ids = inst_probabilities.shape[-1] - 1                                            # column 0 is stuff (:27)
        labels = sorted(torch.unique(labels_gt[things]).cpu().tolist())[:n_ids]            # :29
        if not labels:
            return torch.zeros_like(labels_gt)
        cost = cost_matrix(inst_probabilities, labels_gt, labels, col0=1)                   # :30-33
        assert (self.outlier_rejection and points_3d is not None) or not self.outlier_rejection, \
            "Outlier rejection requires 3d points"                                          # :36-37
        if self.outlier_rejection:                                                           # :38-43
            s, c = label_sums(points_3d.float(), labels_gt, labels)
            cost = id_range_cost(cost, s[:, 0] / c.float())
        rows, cols = scipy.optimize.linear_sum_assignment(np.nan_to_num(cost))              # :45
        # things rays: assigned column + 1; things whose label got no column keep 0 + 1 (:47-53)
        new = _lookup(labels_gt, [labels[r] for r in rows], [int(c) + 1 for c in cols], 1)
        return torch.where(things, new, torch.zeros_like(labels_gt))

    # ---- one host synchronisation per step ----------------------------------------------------------------------------------------------------
    # The general path below asks the device for the sorted unique ids, then for the cost matrix (two synchronisations per IMAGE), and forms the loss from a
    # dozen small tensor ops: ~0.9 ms on a 3.5 ms train step, almost all of it host latency with the GPU idle.  Here every image's cost matrix is built for the
    # image's own sorted distinct ids by three launches (pag_assign_cost), ONE fixed-size copy brings all of them to pinned memory, SciPy runs per image exactly
    # as in the reference (:45), the assigned columns go back in one copy, and the loss is one autograd node (_AssignNLL).  Same arithmetic (fp32
    # sums in ray order, fp32 division, the same -log) - tests/test_gpu_loss.py compares both paths and the reference's golden vectors.
    def _workspace(self, B, P, I, dev):
        key = (B, P, I, str(dev))
        if self._ws is None or self._ws["key"] != key:
            C = R = I - 1                # at most I - 1 labels take part (:29)
            f = dict(key=key,
                     sums=torch.empty(B, R, C, device=dev), counts=torch.empty(B, R, device=dev, dtype=torch.int32),
                     info=torch.zeros(B, 2, device=dev, dtype=torch.int32), labels=torch.zeros(B, R, device=dev, dtype=torch.int64),
                     cost=torch.zeros(B, R, C, device=dev), targets=torch.ones(B, R, device=dev, dtype=torch.int64),
                     psums=torch.zeros(B, R, 3, device=dev), pcounts=torch.zeros(B, R, device=dev, dtype=torch.int32),
                     lo_hi=torch.zeros(B, R, 2, device=dev, dtype=torch.int32))
            f["status"] = torch.zeros(B, device=dev, dtype=torch.int32)
            for name in ("info", "cost", "targets", "lo_hi", "status"):
                f["h_" + name] = torch.empty(f[name].shape, dtype=f[name].dtype).pin_memory()
            self._ws = f
        return self._ws

    DEVICE_SOLVE_MAX = 256       # rows / columns pag_assign_solve takes (BUP20: 199)

    def _device_solver(self, I):
        return self.solver == "device" and I - 1 <= self.DEVICE_SOLVE_MAX

    def _check_last_status(self, w):
        """The device solver reports per image 0 = solved, 1 = more distinct ids than the device-side set holds, 2 = infeasible matrix.  The report of the
        PREVIOUS call is read here, without waiting, once its copy has landed: anything but 0 switches this object to the host solver (whose general path
        covers those inputs) and says so - the call it belongs to trained that image's instance term against all-ones targets."""
        ev = w.get("status_event")
        if ev is not None and ev.query():
            w["status_event"] = None
            bad = w["h_status"].numpy()
            if bad.any():
                import warnings
                warnings.warn("LinAssignmentThingsLoss: the device assignment was not solved for image(s) %s of an earlier step (status %s: 1 = more distinct ids "
                              "than the device-side set holds, 2 = infeasible cost matrix); switching to solver='scipy'" % (np.nonzero(bad)[0].tolist(), bad[bad != 0].tolist()))
                self.solver = "scipy"

    def _fast(self, prob, labels_gt, stuff_mask, points_3d=None):
        return self._finish(self._begin(prob, labels_gt, stuff_mask, points_3d))

    def _begin(self, prob, labels_gt, stuff_mask, points_3d=None, side=False):
        """Queue the device side of the assignment.  solver="device": ids, sums, cost rows, id ranges (pag_assign_cost) and the Hungarian step itself
        (pag_assign_solve) - nothing for the host to wait for; with side=True (the two-call form) on a SECOND STREAM that waits for what the caller has queued so
        far, so that the latency-bound solve (one wave per image) runs beside whatever the caller queues next on its own stream (the colour / density / main-grid
        half of the backward); _finish() makes the caller's stream wait for it.  solver="scipy": pag_assign_cost, the copies to pinned memory and an EVENT behind
        them - the host waits for that event and solves the assignments in _finish()."""
        B, P, I = prob.shape
        w = self._workspace(B, P, I, prob.device)
        # ONE pending call per loss object: the workspace (cost rows, targets, pinned mirrors, the event) belongs to it until finish() has run; a second
        # begin() - two micro-batches in flight, one object shared by two heads - would overwrite the first call's rows without any error
        if w.get("busy"):
            raise RuntimeError("LinAssignmentThingsLoss.begin() called again before finish() of the previous call: one pending call per loss object "
                               "(use one LinAssignmentThingsLoss per concurrently pending batch)")
        w["busy"] = True
        pd = prob.detach()
        pts, slope, x_limit, margin = None, 0.0, 0.0, 0
        names = ("info", "cost")
        if points_3d is not None:
            # outlier rejection (:38-43): the per-id centres and the id range each may take come out of the same launches (utils/outlier_rejection.py:8-51 with its
            # default frame_min_length 0.3, max_num_inst_at_x 30, id_margin 30 - the python scalars the tensor ops would cast to fp32)
            pts = points_3d.detach().float().contiguous()
            margin = 30
            slope = (30 + margin) / 0.3
            x_limit = ((I - 1) - margin) / slope
            names = ("info", "cost", "lo_hi")
        if self._device_solver(I):
            self._check_last_status(w)
        on_device = self._device_solver(I)
        main = torch.cuda.current_stream(prob.device)
        other = None
        if on_device and side:
            other = w.get("side_stream")
            if other is None:
                # (normal priority: with a high-priority stream the 0.3 ms one-wave-per-image solve took dispatch precedence over the backward it runs beside
                # and the two-call step got slower, 3.49 -> 4.29 ms at 4096 rays)
                other = w["side_stream"] = torch.cuda.Stream(device=prob.device)
            other.wait_stream(main)                       # the probabilities, the gt ids and the 3-D points are produced on the caller's stream
        with torch.cuda.stream(other if other is not None else main):
            st = L.stream()
            ops._call("pag_assign_cost", pd.data_ptr(), B, P, pd.stride(0), pd.stride(1), I, 1, labels_gt.data_ptr(), I - 1, w["sums"].data_ptr(), w["counts"].data_ptr(),
                      w["info"].data_ptr(), w["labels"].data_ptr(), w["cost"].data_ptr(), pts.data_ptr() if pts is not None else None, slope, x_limit, margin,
                      w["psums"].data_ptr(), w["pcounts"].data_ptr(), w["lo_hi"].data_ptr(), st)          # every image of the step in one set of launches
            if on_device:
                # the Hungarian step on the device: targets are written where pag_assign_nll_fwd reads them - nothing to copy, nothing for the host to wait for
                ops._call("pag_assign_solve", w["cost"].data_ptr(), B, I - 1, I - 1, w["info"].data_ptr(), w["lo_hi"].data_ptr() if pts is not None else None,
                          w["targets"].data_ptr(), w["status"].data_ptr(), st)
                if w.get("status_event") is None:             # one report in flight at a time (the pinned mirror is single-buffered)
                    w["h_status"].copy_(w["status"], non_blocking=True)
                    sev = w.get("status_event_obj")
                    if sev is None:
                        sev = w["status_event_obj"] = torch.cuda.Event()
                    sev.record(torch.cuda.current_stream(prob.device))
                    w["status_event"] = sev
                done = None
                if other is not None:
                    done = w.get("side_done")
                    if done is None:
                        done = w["side_done"] = torch.cuda.Event()
                    done.record(other)
                return (prob, labels_gt, stuff_mask, points_3d is not None, w, pts, True, done)
        for name in names:
            w["h_" + name].copy_(w[name], non_blocking=True)
        ev = w.get("event")
        if ev is None:
            ev = w["event"] = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(prob.device))
        return (prob, labels_gt, stuff_mask, points_3d is not None, w, pts, False, None)

    def _finish(self, pending):
        prob, labels_gt, stuff_mask, rej, w, _pts, on_device, side_done = pending
        B, P, I = prob.shape
        if on_device:
            w["busy"] = False
            if side_done is not None:
                torch.cuda.current_stream(prob.device).wait_event(side_done)      # a wait of the STREAM: the host queues on
            loss, virt = _AssignNLL.apply(prob, labels_gt, stuff_mask, w["labels"], w["targets"], w["info"])
            self.last_virtual_labels = virt
            return loss
        w["event"].synchronize()                                       # the step's one wait for the device: only for what _begin() queued
        w["busy"] = False                                              # (the targets below are written and consumed inside this call)
        info = w["h_info"].numpy()
        if info[:, 1].any():
            return None                                                # more distinct ids than the device-side set holds: general path
        tg = w["h_targets"].numpy()
        tg.fill(1)                                                     # things whose id got no column keep 0 + 1 (:47-53)
        for b in range(B):
            n = int(info[b, 0])
            if n == 0:
                continue
            cost = w["h_cost"].numpy()[b, :n].astype(np.float64)
            if rej:
                lh = w["h_lo_hi"].numpy()[b, :n]
                ids = np.arange(I - 1)[None, :]
                cost[~((lh[:, :1] <= ids) & (ids <= lh[:, 1:]))] = 10000                        # utils/outlier_rejection.py:8-51
            rows, cols = scipy.optimize.linear_sum_assignment(np.nan_to_num(cost))              # :45
            tg[b, rows] = cols + 1
        w["targets"].copy_(w["h_targets"], non_blocking=True)
        loss, virt = _AssignNLL.apply(prob, labels_gt, stuff_mask, w["labels"], w["targets"], w["info"])
        self.last_virtual_labels = virt       # i64 [B, P]: the virtual ground truth of :23-54 this step trained against (defined on valid rays: stuff | id > 0)
        return loss

    # ---- two-call form (this build's addition) ---------------------------------------------------------------------------------------------------
    def begin(self, inst_probabilities, labels_gt, stuff_mask, points_3d=None):
        """forward() split in two so that device work can be queued between the launches and the host's wait:

            pending = loss_fn.begin(inst, ids, stuff, points_3d)     # pag_assign_cost + copies + event; returns at once
            ...queue anything that does not need the instance loss (other loss terms; `rgb_loss.backward()`: the colour / density / main-grid
               half of the backward - see INTEGRATION.md)...
            inst_loss = loss_fn.finish(pending)                      # waits for the event, SciPy per image (:45), pag_assign_nll_fwd

        finish(begin(...)) == forward(...) (same launches, same values).  ONE pending call per loss object: begin() raises when the previous
        call has not been finished (its workspace would be overwritten); a new workspace shape (another B / P / I) drops a pending call's state."""
        fast = self._gate(inst_probabilities, labels_gt, stuff_mask, points_3d)
        if fast is None:
            return ("general", inst_probabilities, labels_gt, stuff_mask, points_3d)
        return ("fast", self._begin(*fast, side=True), inst_probabilities, labels_gt, stuff_mask, points_3d)

    def finish(self, pending):
        if pending[0] == "fast":
            out = self._finish(pending[1])
            if out is not None:
                return out
            pending = ("general",) + pending[2:]
        _, p3, gt, sm, pts = pending
        return self._general(p3, gt, sm, pts)

    def _gate(self, p3, labels_gt, stuff_mask, points_3d):
        """Arguments of the one-synchronisation path, or None when the inputs are outside what its launches read through raw pointers."""
        assert (self.outlier_rejection and points_3d is not None) or not self.outlier_rejection, "Outlier rejection requires 3d points"      # :36-37
        if not (self.fast_path and torch.is_tensor(p3) and p3.is_cuda and p3.dim() == 3 and p3.dtype == torch.float32
                and p3.stride(2) == 1 and 2 <= p3.shape[2] <= 1025 and torch.is_tensor(labels_gt) and labels_gt.dtype == torch.int64
                and labels_gt.shape == p3.shape[:2] and torch.is_tensor(stuff_mask) and stuff_mask.shape == p3.shape[:2]
                and labels_gt.device == p3.device and stuff_mask.device == p3.device):
            return None
        # (everything the launches read through raw pointers lives on the probabilities' device and has the shape the kernels index with;
        # anything else - a CPU label tensor, a ragged list of points - takes the general path, which raises Python errors)
        pts = None
        if self.outlier_rejection:
            pts = points_3d if torch.is_tensor(points_3d) else torch.stack(list(points_3d))
            if not (pts.device == p3.device and tuple(pts.shape) == (p3.shape[0], p3.shape[1], 3)):
                return None
        gt_c = labels_gt.contiguous()
        sm_c = stuff_mask.contiguous()
        # any non-zero entry is True, as torch.logical_or (:60) reads it (a plain .to(uint8) would wrap 256 to 0 and truncate 0.5)
        sm_c = sm_c.view(torch.uint8) if sm_c.dtype == torch.bool else (sm_c != 0).view(torch.uint8)
        return p3, gt_c, sm_c, pts

    def forward(self, inst_probabilities, labels_gt, stuff_mask, points_3d=None, *args, **kwargs):
        fast = self._gate(inst_probabilities, labels_gt, stuff_mask, points_3d)
        if fast is not None:
            out = self._fast(*fast)
            if out is not None:
                return out
        return self._general(inst_probabilities, labels_gt, stuff_mask, points_3d)

    def _general(self, inst_probabilities, labels_gt, stuff_mask, points_3d=None):
        loss = []
        for i, (p, gt, m) in enumerate(zip(inst_probabilities, labels_gt, stuff_mask)):
            valid = torch.logical_or(m, gt > 0)                                             # :60
            gt_v = torch.where(valid, gt, torch.zeros_like(gt))
            virt = self.create_virtual_gt_with_linear_assignment(p, gt_v, points_3d[i] if points_3d is not None else None)
            wrong = ((virt != p.argmax(dim=-1)) & valid).any()                              # :69,:79
            nll = -torch.log(p.gather(1, virt[:, None])[:, 0] + 1e-27)                      # :80
            loss.append(torch.where(valid & wrong, nll, torch.zeros_like(nll)))
        return torch.stack(loss)


SEGMENT_SLOTS = 2048        # distinct ground-truth ids per image segment_consistency_regularizer() has room for (more: the result is NaN)


SEGMENT_KERNELS = os.environ.get("PAG_SEGMENT_KERNELS", "1") != "0"      # False / PAG_SEGMENT_KERNELS=0: the tensor-op form below on GPU tensors too (tests, A/B)


class _SegmentReg(torch.autograd.Function):
    """pag_segment_reg_fwd / _bwd (csrc/regularizer.hip): the regulariser in four launches forward and one backward."""

    @staticmethod
    def forward(ctx, prob, labels, eps):
        B, P, I = prob.shape
        dev = prob.device
        nbytes = int(L.load().pag_segment_reg_workspace_bytes(B, P))
        ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
        out = torch.empty(1, device=dev)
        ops._call("pag_segment_reg_fwd", prob.data_ptr(), B, P, prob.stride(0), prob.stride(1), I, float(eps), labels.data_ptr(), ws.data_ptr(), nbytes,
                  out.data_ptr(), L.stream())
        ctx.save_for_backward(prob, ws)
        ctx.eps = float(eps)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        prob, ws = ctx.saved_tensors
        B, P, I = prob.shape
        g = g.reshape(1).contiguous().float()
        d = torch.empty(B, P, I, device=prob.device)
        ops._call("pag_segment_reg_bwd", prob.data_ptr(), B, P, prob.stride(0), prob.stride(1), I, ctx.eps, ws.data_ptr(), ws.numel(), g.data_ptr(), d.data_ptr(),
                  L.stream())
        return d, None, None


def segment_consistency_regularizer(embeddings, labels, eps=0.0):
    """loss/regularizers.py:5-35 on device tensors WITHOUT a host synchronisation: `embeddings` [B,P,I] probabilities, `labels` int64 [B,P] -> 0-dim tensor,
    differentiable with respect to `embeddings`.  The reference's caller adds 1e-27 to the probabilities first (pc_nerf/trainer.py:525-527:
    `segment_consistency_regularizer((inst_embed + 1e-27).reshape(B, -1, I), ...)`); pass the bare probabilities and `eps=1e-27` instead and the kernels add
    it as they read (one [B,P,I] temporary and its backward less; same fp32 sums).

    fp32 CUDA tensors take pag_segment_reg_fwd / _bwd (four launches + one; every sum in a fixed order); anything else the tensor-op form below.

    The reference loops over images and segments on the host (unique -> .cpu() -> tensor_split, one bincount / arg-max / nll_loss per segment: a dozen
    device round trips per segment).  Same quantities here from ~25 launches over the whole batch: the rays' segment slot = rank of their id among the
    image's sorted distinct ids - EVERY distinct value is a segment, 0 included (:11-18); per (image, slot) the histogram of the rays' arg-max column (:22);
    slots whose rays all predict column 0 are skipped (:24-25); label = first most frequent column among 1.. (:27), 0 when bins[0] * 0.5 > bins[label]
    (:29-30); term = mean over the slot's rays of -log p[ray, label] (:32); after each image the RUNNING total is divided by that image's number of
    segments (:33); finally by the number of images (:35).  Values agree with the reference up to the fp32 summation order of the per-segment means
    (tests/test_gpu_loss.py against tests/golden/g6_reg.npz: value and gradient of the reference function)."""
    B, P, I = embeddings.shape
    if SEGMENT_KERNELS and embeddings.is_cuda and embeddings.dtype == torch.float32 and labels.device == embeddings.device and B >= 1 and P >= 1 and I <= 4096:
        prob = embeddings if embeddings.stride(-1) == 1 else embeddings.contiguous()
        return _SegmentReg.apply(prob, labels.detach().reshape(B, P).contiguous().long(), eps)
    if eps:
        embeddings = embeddings + eps
    S = min(P, SEGMENT_SLOTS)
    arg = embeddings.detach().argmax(-1)                                                   # [B,P]  :22
    srt, order = torch.sort(labels, dim=1)                                                 # :11-12
    new = torch.ones_like(srt, dtype=torch.bool)
    new[:, 1:] = srt[:, 1:] != srt[:, :-1]
    slot_sorted = torch.cumsum(new, dim=1) - 1
    n_seg = slot_sorted[:, -1] + 1                                                         # [B] distinct ids per image
    slot = torch.empty_like(slot_sorted).scatter_(1, order, slot_sorted).clamp_(max=S - 1)  # [B,P] slot of each ray
    hist = torch.zeros(B, S * I, dtype=torch.int32, device=embeddings.device)
    hist.scatter_add_(1, slot * I + arg, torch.ones(1, dtype=torch.int32, device=embeddings.device).expand(B, P))
    hist = hist.view(B, S, I)
    rest = hist[..., 1:]
    best = rest.argmax(-1) + 1                                                             # :27 (ties: the lowest column, as bincount().argmax())
    skipped = rest.sum(-1) == 0                                                            # :24-25 (and the slots no ray maps to)
    b0 = hist[..., 0]
    bb = hist.gather(-1, best[..., None])[..., 0]
    best = torch.where(b0 * 0.5 > bb, torch.zeros_like(best), best)                        # :29-30
    lab_ray = best.gather(1, slot)                                                         # [B,P]
    use = ~skipped.gather(1, slot)
    nll = -torch.log(embeddings.gather(-1, lab_ray[..., None])[..., 0])                    # :32
    per_seg = torch.zeros(B, S, dtype=nll.dtype, device=nll.device).scatter_add(1, slot, torch.where(use, nll, torch.zeros_like(nll)))
    per_img = (per_seg / hist.sum(-1).clamp(min=1)).sum(1)                                 # segment means, summed per image
    reg = per_img.new_zeros(())
    for b in range(B):
        reg = (reg + per_img[b]) / n_seg[b]                                                # :33 - the running total, earlier images included
    reg = reg / B                                                                          # :35
    return torch.where((n_seg > S).any(), torch.full_like(reg, float("nan")), reg)         # more ids than slots: loud, not wrong


class LinAssignmentLoss(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()

    @torch.no_grad()
    def create_virtual_gt_with_linear_assignment(self, labels_gt, predicted_scores):
        """:16-26 (softmax of the scores inside, every gt label takes part)."""
        labels = sorted(torch.unique(labels_gt).cpu().tolist())[:predicted_scores.shape[-1]]
        prob = torch.softmax(predicted_scores, dim=-1)
        cost = cost_matrix(prob, labels_gt, labels)
        rows, cols = scipy.optimize.linear_sum_assignment(np.nan_to_num(cost))
        return _lookup(labels_gt, [labels[r] for r in rows], [int(c) for c in cols], 0)

    def forward(self, inst_embeddings, labels_gt, *args, **kwargs):
        loss = torch.zeros(1, device=labels_gt.device)
        for s, gt in zip(inst_embeddings, labels_gt):
            virt = self.create_virtual_gt_with_linear_assignment(gt, s)
            wrong = (virt != s.argmax(dim=-1)).any()                                        # :32
            nll = -torch.log(s.gather(1, virt[:, None])[:, 0] + 1e-27).mean()               # :33
            loss = loss + torch.where(wrong, nll, torch.zeros_like(nll))
        return loss / inst_embeddings.shape[0]
