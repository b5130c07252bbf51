"""Oracle: the loss-side regularisers that consume the hot path's outputs, numpy.

TEST INFRASTRUCTURE - see oracle/__init__.py.

Follows:
  loss/regularizers.py:5-35    segment_consistency_regularizer(embeddings [B,P,I], labels [B,P])
  loss/regularizers.py:37-39   sigma_sparsity_loss
  utils/outlier_rejection.py:74-97   rays_to_3d_points (the unprojection that feeds the outlier rejection; the camera
                                     transform itself is kaolin's inv_transform_rays - third party, restated as
                                     R^T (o - t) / R^T d in pagnerf_amd/ba_pipeline.py, PARITY UNPINNED)

Called by the reference trainer as `segment_consistency_regularizer((inst_embed + 1e-27).reshape(B, -1, I), inst_gts.reshape(B, -1))`
(pc_nerf/trainer.py:525-527; active in configs/bup20/best.yaml because trainer.py:93 assigns the WEIGHT 1.0 to
`inst_segment_reg_epoch_start`).  Pinned by tests/golden/g6_reg.npz, generated from the reference function itself (value and
autograd gradient).

What the function computes, quirks included:
  per image, the rays are grouped by ground-truth id - EVERY distinct value of `labels` is a segment, id 0 included (:11-18;
  the "excluding the stuff/bg segment" of its comment is not what `tensor_split(sample_idxs[1:])` does);
  per segment: histogram of the rays' arg-max column (:22); a segment whose rays all predict column 0 is skipped (:24-25);
  else the most frequent column among 1.. (first maximum) is the segment's label (:27), replaced by 0 when more than twice as
  many rays predict 0 (:29-30); the term is the mean over the segment's rays of -log(prob[ray, label]) (:32);
  after an image's segments the RUNNING total - earlier images included - is divided by that image's number of segments (:33);
  the result is the total divided by the number of images (:35).
"""
import numpy as np


def segment_labels(prob, labels):
    """-> list of (segment id, ray mask, chosen column or None when the segment is skipped) for ONE image (:11-30)."""
    out = []
    for u in np.unique(labels):                          # sorted, as torch.unique / argsort order the segments (:11-12)
        m = labels == u
        bins = np.bincount(prob[m].argmax(-1))           # :22 (first maximum wins a tie, as torch.argmax on CPU)
        if bins[1:].size == 0:                           # :24-25
            out.append((u, m, None))
            continue
        best = int(bins[1:].argmax()) + 1                # :27
        if bins[0] * 0.5 > bins[best]:                   # :29-30
            best = 0
        out.append((u, m, best))
    return out


def segment_consistency_regularizer(embeddings, labels, want_grad=False):
    """embeddings f32 [B,P,I] (probabilities, already + 1e-27 by the caller), labels int [B,P] -> f32 scalar
    (and d / d embeddings f32 [B,P,I] with want_grad)."""
    embeddings = np.asarray(embeddings, dtype=np.float32)
    B = embeddings.shape[0]
    reg = np.float32(0.0)
    per_image = []
    for x, l in zip(embeddings, labels):
        segs = segment_labels(x, l)
        for _, m, best in segs:
            if best is not None:
                reg = np.float32(reg + np.mean(-np.log(x[m][:, best]), dtype=np.float32))       # :32
        reg = np.float32(reg / np.float32(len(segs)))                                            # :33 - the running total
        per_image.append(segs)
    value = np.float32(reg / np.float32(B))                                                      # :35
    if not want_grad:
        return value
    grad = np.zeros_like(embeddings)
    scale = 1.0 / B
    for b in reversed(range(B)):                          # image b's terms are divided by every LATER image's segment count too
        scale /= len(per_image[b])
        for _, m, best in per_image[b]:
            if best is not None:
                rows = np.nonzero(m)[0]
                grad[b, rows, best] = -scale / (len(rows) * embeddings[b, rows, best])
    return value, grad


def sigma_sparsity_loss(sigma):
    """loss/regularizers.py:37-39 - Cauchy sparsity on the densities."""
    sigma = np.asarray(sigma, dtype=np.float32)
    return np.log(np.float32(1.0) + np.float32(2.0) * sigma * sigma)


def rays_to_3d_points(origins_c, dirs_c, depth, R, t):
    """utils/outlier_rejection.py:74-97 for ONE camera with world->camera rotation R [3,3] and translation t [3]:
    points_cam = dirs * depth (:89); (o_w, p_w) = inv_transform_rays(origins, points_cam) = (R^T (o - t), R^T p) (:91, kaolin -
    restated); points = o_w + p_w (:93).  origins_c / dirs_c [n,3] camera frame, depth [n] -> [n,3] world."""
    p_cam = dirs_c * depth[:, None]
    return (origins_c - t[None]) @ R + p_cam @ R
