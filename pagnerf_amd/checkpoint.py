"""Importing a checkpoint written by the reference (config_parser.py:753-776: `torch.save(pipeline)` /
`pipeline.state_dict()`) into the HIP pipeline.

What maps one to one (same module / parameter names): the four (five) decoders `nef.decoder_*.layers.N.{weight,bias}`,
`nef.decoder_*.lout.{weight,bias}`, BAPipeline's `camera_extrinsics`.  What needs a conversion:

  * grid tables - the reference keeps them inside third-party encoder modules (`permutohedral_encoding.PermutoEncoding`,
    `HashEmbedder.embeddings[i].weight` in grids/hash_grid_torch.py:61-65); they are located by SHAPE under the
    `nef.grid.` / `nef.delta_grid.` prefixes: one `[L, capacity, F]` tensor, or L tensors `[2^log2T, F]` in level order;
    a `[L, 3]` tensor is the permutohedral per-level shift;
  * occupancy - wisp's OctreeAS buffers (`blas_octree`, grids/permuto_grid.py:33-38): kaolin's SPC octree is one byte per
    node in breadth-first order, bit i of a byte = child i present, children in Morton order i = 4x + 2y + z (recalled from
    the public kaolin documentation; third party, PARITY UNPINNED).  octree_to_bits() expands it into the dense bitfield
    the march kernels read.
"""
import torch


def octree_to_bits(octree, level):
    """uint8 [n_nodes] breadth-first SPC octree -> int32 [ceil(8^level / 32)] occupancy bitfield, bit (x*R + y)*R + z."""
    octree = octree.detach().cpu().to(torch.uint8).long()
    codes = torch.zeros(1, dtype=torch.long)                 # Morton codes of the nodes of the current level
    pos = 0
    bit = torch.arange(8)
    for _ in range(level):
        n = codes.shape[0]
        node_bytes = octree[pos:pos + n]
        assert node_bytes.shape[0] == n, "octree shorter than its own hierarchy"
        pos += n
        present = ((node_bytes[:, None] >> bit[None, :]) & 1).bool()          # [n, 8]
        codes = (codes[:, None] * 8 + bit[None, :])[present]
    R = 2 ** level
    x = torch.zeros_like(codes)
    y = torch.zeros_like(codes)
    z = torch.zeros_like(codes)
    for b in range(level):                                   # de-interleave: code = ... x_b y_b z_b ... (x highest of each triple)
        x |= ((codes >> (3 * b + 2)) & 1) << b
        y |= ((codes >> (3 * b + 1)) & 1) << b
        z |= ((codes >> (3 * b)) & 1) << b
    mask = torch.zeros(R ** 3, dtype=torch.bool)
    mask[(x * R + y) * R + z] = True
    return mask_to_bits(mask)


def bits_to_octree(bits, level):
    """Inverse of octree_to_bits (used by the tests and to write reference-readable checkpoints)."""
    R = 2 ** level
    words = bits.detach().cpu().long() & 0xFFFFFFFF
    mask = ((words[:, None] >> torch.arange(32)) & 1).bool().reshape(-1)[:R ** 3].reshape(R, R, R)
    xs, ys, zs = torch.nonzero(mask, as_tuple=True)
    codes = torch.zeros_like(xs)
    for b in range(level):
        codes |= (((xs >> b) & 1) << (3 * b + 2)) | (((ys >> b) & 1) << (3 * b + 1)) | (((zs >> b) & 1) << (3 * b))
    levels = []
    cur = torch.unique(codes)
    for _ in range(level):
        parents, inv = torch.unique(cur >> 3, return_inverse=True)
        byte = torch.zeros(parents.shape[0], dtype=torch.long)
        byte.scatter_add_(0, inv, 1 << (cur & 7))
        levels.append(byte.to(torch.uint8))
        cur = parents
    return torch.cat(list(reversed(levels))) if levels else torch.zeros(0, dtype=torch.uint8)


def mask_to_bits(mask):
    mask = mask.reshape(-1).bool()
    pad = (-mask.numel()) % 32
    if pad:
        mask = torch.cat([mask, mask.new_zeros(pad)])
    w = (mask.reshape(-1, 32).long() << torch.arange(32)).sum(1)
    return torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32)


def _grid_tensors(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix) and isinstance(v, torch.Tensor)}


def load_reference_state_dict(pipeline, state_dict, strict_decoders=True):
    """Copy a reference state_dict (or pickled pipeline's .state_dict()) into `pipeline` (Pipeline / BAPipeline with a
    PanopticDeltaNeF).  Returns the list of reference keys that were not used."""
    sd = state_dict.state_dict() if hasattr(state_dict, "state_dict") else dict(state_dict)
    used = set()
    nef = pipeline.nef
    own = nef.state_dict()
    with torch.no_grad():
        for k, v in sd.items():                                          # decoders, camera_extrinsics: same names
            kk = k[len("nef."):] if k.startswith("nef.") else None
            if kk is not None and "decoder" in kk and kk in own:
                if own[kk].shape != v.shape:
                    if strict_decoders:
                        raise RuntimeError("shape mismatch for %s: %s vs %s" % (k, tuple(v.shape), tuple(own[kk].shape)))
                    continue
                own[kk].copy_(v)
                used.add(k)
            elif k == "camera_extrinsics" and hasattr(pipeline, "camera_extrinsics") and pipeline.camera_extrinsics.shape == v.shape:
                pipeline.camera_extrinsics.copy_(v)
                used.add(k)
        for name in ("grid", "delta_grid"):
            grid = getattr(nef, name, None)
            if grid is None:
                continue
            prefix = "nef.%s." % name
            parts = _grid_tensors(sd, prefix)
            L, T, F = grid.tables.shape
            whole = [k for k, v in parts.items() if tuple(v.shape) == (L, T, F)]
            per_level = sorted([k for k, v in parts.items() if tuple(v.shape) == (T, F)],
                               key=lambda s: [int(t) if t.isdigit() else t for t in s.replace(".", " ").split()])
            if whole:
                grid.tables.copy_(parts[whole[0]].to(grid.tables.dtype))
                used.add(prefix + whole[0])
            elif len(per_level) == L:
                grid.tables.copy_(torch.stack([parts[k] for k in per_level]).to(grid.tables.dtype))
                used.update(prefix + k for k in per_level)
            shift = [k for k, v in parts.items() if tuple(v.shape) == (L, 3)]
            if shift and hasattr(grid, "random_shift_per_level"):
                grid.init_from_scales(random_shift=parts[shift[0]].float(), tables=grid.tables.detach())
                used.add(prefix + shift[0])
            if "blas_octree" in parts:
                grid.blas_init_bits(octree_to_bits(parts["blas_octree"], grid.blas_level).to(grid.blas_bits.device))
                used.update(prefix + k for k in parts if k.startswith("blas_"))
    return sorted(set(sd) - used)
