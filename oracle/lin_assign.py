"""Oracle: linear-assignment virtual instance labels (consumer of the hot path's
``inst_embedding`` output), numpy + SciPy.

TEST INFRASTRUCTURE - see oracle/__init__.py.

Follows:
  loss/lin_assignment.py:16-26           cost = -mean prob of each gt label, Hungarian, relabel
  loss/lin_assignment_things.py:23-54    things-only variant (label 0 = stuff; column 0 dropped;
                                         result shifted by +1), optional ID-range cost
  loss/lin_assignment_things.py:56-82    per-image masking (stuff mask | gt>0) and NLL
  utils/outlier_rejection.py:8-51        add_position_id_range_cost
  utils/outlier_rejection.py:56-71       centers_from_3d_points_with_ids

north_star requires the assignment indices to be bit-exact; tests compare against
tests/golden/g5_linassign.npz generated from the reference files above.
"""
import numpy as np
from scipy.optimize import linear_sum_assignment


def _cost_matrix(prob, gt, labels):
    cost = np.zeros([len(labels), prob.shape[-1]])
    for li, lab in enumerate(labels):
        m = gt == lab
        # fp32 sum / (count + 1e-4) exactly as torch does it, then negated into float64
        s = prob[m].sum(axis=0, dtype=np.float32)
        cost[li, :] = -(s / np.float32(m.sum() + 1e-4))
    return cost


def virtual_labels(prob, gt):
    """loss/lin_assignment.py:16-26 with probabilities already soft-maxed by the caller.
    prob f32 [P,I], gt int64 [P] -> int64 [P]."""
    labels = sorted(np.unique(gt).tolist())[:prob.shape[-1]]
    cost = _cost_matrix(prob, gt, labels)
    rows, cols = linear_sum_assignment(np.nan_to_num(cost))
    new = np.zeros_like(gt)
    for a, li in enumerate(rows):
        new[gt == labels[li]] = cols[a]
    return new


def centers_from_points(points_ids):
    """[P,4] (x,y,z,id) -> [K,4] mean position per id (fp32)."""
    ids = np.unique(points_ids[:, 3])
    same = (ids[:, None] == points_ids[None, :, 3]).astype(np.float32)
    centers = (same @ points_ids[:, :3].astype(np.float32)) / same.sum(axis=1)[:, None]
    return np.concatenate([centers, ids[:, None].astype(np.float32)], axis=1)


def id_range_cost(cost, centers, frame_min_length=0.3, max_num_inst_at_x=30, id_margin=30):
    num_ids = cost.shape[1]
    m = np.float32((max_num_inst_at_x + id_margin) / frame_min_length)
    x_limit = np.float32((num_ids - id_margin) / ((max_num_inst_at_x + id_margin) / frame_min_length))
    x = ((-centers[:, 0].astype(np.float32) + np.float32(1)) / np.float32(2)).astype(np.float32)
    lo = np.clip(m * np.mod(x, x_limit), 0, num_ids - 1).astype(np.int64)
    hi = np.clip(lo + id_margin, 0, num_ids - 1)
    ar = np.arange(num_ids)[None, :]
    ok = (lo[:, None] <= ar) & (ar <= hi[:, None])
    cost = cost.copy()
    cost[~ok] = 10000
    return cost


def virtual_labels_things(prob, gt, points_3d=None, outlier_rejection=False):
    """loss/lin_assignment_things.py:23-54.  prob f32 [P,I], gt int64 [P] -> int64 [P]."""
    things = gt > 0
    tgt = gt[things]
    tprob = prob[things][..., 1:]
    labels = sorted(np.unique(tgt).tolist())[:tprob.shape[-1]]
    cost = _cost_matrix(tprob, tgt, labels)
    if outlier_rejection:
        pts = np.concatenate([points_3d[things], tgt[:, None].astype(points_3d.dtype)], axis=-1)
        cost = id_range_cost(cost, centers_from_points(pts))
    rows, cols = linear_sum_assignment(np.nan_to_num(cost))
    tl = np.zeros_like(tgt)
    for a, li in enumerate(rows):
        tl[tgt == labels[li]] = cols[a]
    new = np.zeros_like(gt)
    new[things] = tl + 1
    return new


def lsap_jv(cost):
    """Restatement of SciPy's `linear_sum_assignment` for a float64 cost matrix with rows <= columns (the only shape the loss produces:
    at most I - 1 labels against I - 1 columns, loss/lin_assignment_things.py:29), sequential, in SciPy's own operation order - the
    specification the device kernel `pag_assign_solve` is checked against, itself PINNED against the SciPy installed here (1.15.3) in
    tests/test_oracle_golden.py on thousands of matrices including tie-heavy integer ones.

    Algorithm of record: scipy/optimize/rectangular_lsap/rectangular_lsap.cpp (SciPy >= 1.6; third-party, not in /root/reference, pinned by
    no requirements file there - requirements.txt names `scipy` without a version): the shortest-augmenting-path form of Jonker-Volgenant
    after D. F. Crouse, "On implementing 2D rectangular assignment algorithms", IEEE T-AES 52(4), 2016.  Per row curRow: a Dijkstra-like
    search over the columns not yet scanned (`remaining`, filled in REVERSE order, removed by swap-with-last) with reduced costs
    minVal + c[i,j] - u[i] - v[j]; among equal shortest-path costs a column WITHOUT a row wins (the last such in scan order); dual update;
    augmentation along `path`.  Returns col4row int64 [rows] (row i -> column)."""
    cost = np.asarray(cost, dtype=np.float64)
    nr, nc = cost.shape
    assert nr <= nc
    u, v = np.zeros(nr), np.zeros(nc)
    path = np.full(nc, -1, dtype=np.int64)
    col4row = np.full(nr, -1, dtype=np.int64)
    row4col = np.full(nc, -1, dtype=np.int64)
    for cur in range(nr):
        spc = np.full(nc, np.inf)
        SR, SC = np.zeros(nr, dtype=bool), np.zeros(nc, dtype=bool)
        remaining = [nc - it - 1 for it in range(nc)]
        n_rem, min_val, i, sink = nc, 0.0, cur, -1
        while sink == -1:
            index, lowest = -1, np.inf
            SR[i] = True
            for it in range(n_rem):
                j = remaining[it]
                r = min_val + cost[i, j] - u[i] - v[j]
                if r < spc[j]:
                    path[j] = i
                    spc[j] = r
                if spc[j] < lowest or (spc[j] == lowest and row4col[j] == -1):
                    lowest = spc[j]
                    index = it
            min_val = lowest
            assert min_val != np.inf, "infeasible cost matrix"
            j = remaining[index]
            if row4col[j] == -1:
                sink = j
            else:
                i = row4col[j]
            SC[j] = True
            n_rem -= 1
            remaining[index] = remaining[n_rem]
        u[cur] += min_val
        for r_ in range(nr):
            if SR[r_] and r_ != cur:
                u[r_] += min_val - spc[col4row[r_]]
        for c_ in range(nc):
            if SC[c_]:
                v[c_] -= min_val - spc[c_]
        j = sink
        while True:
            i = path[j]
            row4col[j] = i
            col4row[i], j = j, col4row[i]
            if i == cur:
                break
    return col4row
