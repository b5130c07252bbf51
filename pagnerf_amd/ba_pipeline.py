"""BAPipeline: nef + tracer + learnable camera extrinsics (pc_nerf/ba_pipeline.py).

The reference keeps the poses in a kaolin `Camera` switched to the 'matrix_6dof_rotation' backend
(:44) and registers its parameter tensor as `camera_extrinsics` (:49-51); per step the canonical
camera-frame `base_rays` of each image are mapped to world space with `inv_transform_rays`, the
directions re-normalised (:85-92), and the tracer runs on the result.  kaolin is a third-party
dependency that is not part of this build, so the representation is restated here (PARITY
UNPINNED, recalled from the public kaolin sources):

  params [C, 9] = (r1[3], r2[3], t[3]);  b1 = r1/|r1|, b2 = normalise(r2 - (b1.r2) b1), b3 = b1 x b2
  view matrix  V = [R | t] with rows of R = (b1, b2, b3)      (world -> camera)
  inv_transform_rays: origins_w = R^T (o_c - t),  dirs_w = R^T d_c

These are per-RAY 3x3 products (N = 24 576 per step at best.yaml sizes).  On the GPU they are one launch
each way (ops.pose_rays: pag_pose_rays_fwd / _bwd - the tensor-op form below is ~25 launches forward and
~60 backward, 0.8 ms of a 3 ms post-prune step); CPU tensors (plumbing tests) take the tensor ops.  What
they make necessary on the hot path is the pose gradient through the packed samples: d loss / d xyz from
the encoders (pag_*_encode_bwd_xyz), the per-ray sums of ops.ray_samples and the view-embedding
gradient of the colour decoder (pag_view_embed_bwd).
"""
import torch
import torch.nn as nn

from . import ops
from .core import Pipeline, Rays


def rotation_6d_to_matrix(r6):
    """[C,6] -> [C,3,3] rotation with rows (b1, b2, b3) (Gram-Schmidt, Zhou et al. 2019)."""
    a1, a2 = r6[:, 0:3], r6[:, 3:6]
    b1 = a1 / torch.linalg.norm(a1, dim=-1, keepdim=True)
    b2 = a2 - (b1 * a2).sum(-1, keepdim=True) * b1
    b2 = b2 / torch.linalg.norm(b2, dim=-1, keepdim=True)
    b3 = torch.cross(b1, b2, dim=-1)
    return torch.stack([b1, b2, b3], dim=-2)


def extrinsics_from_view_matrix(view):
    """[C,4,4] world->camera matrices -> [C,9] parameters."""
    return torch.cat([view[:, 0, :3], view[:, 1, :3], view[:, :3, 3]], dim=-1).float()


class BAPipeline(Pipeline):
    def __init__(self, nef, view_matrices, tracer=None, anchor_frame_idxs=(), pose_opt_only_frame_idxs=(), cam_ids=None,
                 near=0.0, far=2.0):
        super().__init__(nef, tracer)
        view_matrices = torch.as_tensor(view_matrices, dtype=torch.float32)
        assert view_matrices.dim() == 3 and view_matrices.shape[0] > 1, "needs more than one camera (ba_pipeline.py:34-37)"
        self.cam_id_to_idx = {cid: i for i, cid in enumerate(cam_ids)} if cam_ids is not None else None      # :29-31
        self.anchor_frame_idxs = list(anchor_frame_idxs)
        self.pose_opt_only_frame_idxs = list(pose_opt_only_frame_idxs)
        self.near, self.far = near, far
        self.camera_extrinsics = nn.Parameter(extrinsics_from_view_matrix(view_matrices))               # :49-51
        self._hooked = False

    def to(self, *args, **kwargs):
        out = super().to(*args, **kwargs)
        if self.anchor_frame_idxs and not self._hooked:                                                  # :56-60
            grad_mask = torch.ones_like(self.camera_extrinsics)
            grad_mask[self.anchor_frame_idxs] = 0.0
            self.camera_extrinsics.register_hook(lambda grad: grad * grad_mask.to(grad.device))
            self._hooked = True
        return out

    def view_matrices(self):
        R = rotation_6d_to_matrix(self.camera_extrinsics[:, :6])
        top = torch.cat([R, self.camera_extrinsics[:, 6:, None]], dim=-1)
        bottom = torch.tensor([0.0, 0.0, 0.0, 1.0], device=top.device).expand(top.shape[0], 1, 4)
        return torch.cat([top, bottom], dim=1)

    def camera_indices(self, cam_ids):
        assert isinstance(cam_ids, (tuple, list, torch.Tensor))                                           # :79
        if isinstance(cam_ids, (tuple, list)):
            if self.cam_id_to_idx is not None:
                cam_ids = [self.cam_id_to_idx[c] for c in cam_ids]
            cam_ids = torch.tensor(cam_ids, dtype=torch.long)
        assert cam_ids.nelement() > 0                                                                     # :82
        return cam_ids.to(self.camera_extrinsics.device)

    def transform_rays(self, base_rays, cam_ids):
        """:85-92 - base rays of len(cam_ids) images, [C*n,3] camera-frame -> world-frame Rays (dirs unit length)."""
        idx = self.camera_indices(cam_ids)
        if self.camera_extrinsics.is_cuda and base_rays.origins.numel() and base_rays.origins.numel() % (3 * len(idx)) == 0:
            dev = self.camera_extrinsics.device
            o, d = base_rays.origins.reshape(-1, 3).to(dev), base_rays.dirs.reshape(-1, 3).to(dev)
            ow, dw = ops.pose_rays(self.camera_extrinsics, idx.int(), o.shape[0] // len(idx), o, d)       # one launch (pag_pose_rays_fwd)
            return Rays(ow, dw, dist_min=self.near, dist_max=self.far)
        prm = self.camera_extrinsics[idx]
        R = rotation_6d_to_matrix(prm[:, :6])                       # [C,3,3] world -> camera
        t = prm[:, 6:]
        o = base_rays.origins.reshape(len(idx), -1, 3).to(prm.device)
        d = base_rays.dirs.reshape(len(idx), -1, 3).to(prm.device)
        origins = torch.matmul(o - t[:, None, :], R)                # row-vector form of R^T (o - t)
        dirs = torch.matmul(d, R)
        dirs = dirs / torch.linalg.norm(dirs, dim=-1, keepdim=True)
        return Rays(origins.float().reshape(-1, 3), dirs.float().reshape(-1, 3), dist_min=self.near, dist_max=self.far)

    def transform_rays_indexed(self, origins_c, dirs_c, cam_idx):
        """Per-ray form of transform_rays(): ray i belongs to camera `cam_idx[i]` (row of camera_extrinsics), so a ray shard of a
        multi-GPU step may start and end in the middle of an image.  The arithmetic of :85-92 (see the note below)."""
        if self.camera_extrinsics.is_cuda and origins_c.numel():
            cam = cam_idx if (cam_idx.dtype == torch.int32 and cam_idx.device == self.camera_extrinsics.device) else \
                cam_idx.to(device=self.camera_extrinsics.device, dtype=torch.int32)
            ow, dw = ops.pose_rays(self.camera_extrinsics, cam, 1, origins_c.reshape(-1, 3), dirs_c.reshape(-1, 3))
            return Rays(ow, dw, dist_min=self.near, dist_max=self.far)
        R_all = rotation_6d_to_matrix(self.camera_extrinsics[:, :6])      # [C,3,3] world -> camera, once per camera
        idx = cam_idx.to(self.camera_extrinsics.device).long()
        R = R_all.index_select(0, idx)                                   # [n,3,3]
        t = self.camera_extrinsics[:, 6:].index_select(0, idx)
        # row-vector form of R^T (o - t) with a matrix PER RAY: written as three scaled rows - n tiny (1x3)(3x3) products through the
        # batched-GEMM library cost 0.2 ms per call and two more each in the backward (1.1 ms of a 27 ms step).  Products and sums in the
        # same order as the dot products of transform_rays(); a library GEMM may fuse them into FMAs (<= 1 ulp apart).
        v = origins_c - t
        origins = v[:, 0:1] * R[:, 0] + v[:, 1:2] * R[:, 1] + v[:, 2:3] * R[:, 2]
        dirs = dirs_c[:, 0:1] * R[:, 0] + dirs_c[:, 1:2] * R[:, 1] + dirs_c[:, 2:3] * R[:, 2]
        dirs = dirs / torch.linalg.norm(dirs, dim=-1, keepdim=True)
        return Rays(origins.float(), dirs.float(), dist_min=self.near, dist_max=self.far)

    @torch.no_grad()
    def rays_to_3d_points(self, base_rays, depth, cam_ids):
        """utils/outlier_rejection.py:74-97 as pc_nerf/trainer.py:508-518 calls it (`rays` there are the camera-frame base rays, `depth`
        the composited depth of the same step, `cameras` the step's current extrinsics): points_cam = dirs * depth (:89), mapped to the
        world by the cameras' inv_transform_rays (:91, the arithmetic of transform_rays above) and added to the transformed origins (:93).
        -> [C*n, 3] world points, no gradient (the trainer wraps the call in torch.no_grad())."""
        idx = self.camera_indices(cam_ids)
        if self.camera_extrinsics.is_cuda and base_rays.origins.numel() and base_rays.origins.shape[0] % len(idx) == 0:
            dev = self.camera_extrinsics.device
            return ops.pose_points(self.camera_extrinsics, idx.int(), base_rays.origins.shape[0] // len(idx), base_rays.origins.reshape(-1, 3).to(dev),
                                   base_rays.dirs.reshape(-1, 3).to(dev), depth.to(dev))
        prm = self.camera_extrinsics[idx]
        R, t = rotation_6d_to_matrix(prm[:, :6]), prm[:, 6:]
        o = base_rays.origins.reshape(len(idx), -1, 3).to(prm.device)
        d = base_rays.dirs.reshape(len(idx), -1, 3).to(prm.device)
        p_cam = d * depth.reshape(len(idx), -1, 1).to(prm.device)
        return (torch.matmul(o - t[:, None, :], R) + torch.matmul(p_cam, R)).float().reshape(-1, 3)

    @torch.no_grad()
    def rays_to_3d_points_indexed(self, origins_c, dirs_c, depth, cam_idx):
        """Per-ray form of rays_to_3d_points() (ray i belongs to camera cam_idx[i]; see transform_rays_indexed)."""
        if self.camera_extrinsics.is_cuda and origins_c.numel():
            cam = cam_idx if (cam_idx.dtype == torch.int32 and cam_idx.device == self.camera_extrinsics.device) else \
                cam_idx.to(device=self.camera_extrinsics.device, dtype=torch.int32)
            return ops.pose_points(self.camera_extrinsics, cam, 1, origins_c.reshape(-1, 3), dirs_c.reshape(-1, 3), depth)
        idx = cam_idx.to(self.camera_extrinsics.device).long()
        R = rotation_6d_to_matrix(self.camera_extrinsics[:, :6]).index_select(0, idx)
        t = self.camera_extrinsics[:, 6:].index_select(0, idx)
        v = origins_c - t + dirs_c * depth.reshape(-1, 1)               # R^T (o - t) + R^T (d * depth) = R^T (o - t + d * depth)
        return (v[:, 0:1] * R[:, 0] + v[:, 1:2] * R[:, 1] + v[:, 2:3] * R[:, 2]).float()

    def forward(self, *args, cam_ids=None, **kwargs):
        if isinstance(cam_ids, (tuple, list, torch.Tensor)):                                              # :69-70
            kwargs["rays"] = self.transform_rays(kwargs["rays"], cam_ids)
        return super().forward(*args, **kwargs)
