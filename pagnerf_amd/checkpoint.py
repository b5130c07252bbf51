"""Checkpoints shared with the reference (config_parser.py:753-776: `torch.save(pipeline)` / `pipeline.state_dict()`): importing one
it wrote into the HIP pipeline (load_reference_state_dict) and writing one it can load (save_reference_state_dict).

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


def spc_buffers(bits, level):
    """The four buffers wisp's OctreeAS keeps beside a grid (grids/permuto_grid.py:33-38, grids/occtree.py:69-74) for the occupancy
    bitfield `bits`: blas_octree (bits_to_octree), blas_points int16 [n_points, 3] (integer coordinates of every node of every level,
    root first, each level in the octree's own breadth-first = Morton order), blas_pyramid int32 [2, level + 2] (row 0: points per level
    followed by 0, row 1: their exclusive prefix sum followed by the total) and blas_prefix int32 [n_bytes] (exclusive prefix sum of the
    child counts of the octree bytes).  Layouts recalled from kaolin's public SPC documentation (kaolin.ops.spc.scan_octrees /
    generate_points): third party, PARITY UNPINNED - load_reference_state_dict() itself needs blas_octree only."""
    octree = bits_to_octree(bits, level)
    counts = torch.tensor([bin(int(b)).count("1") for b in octree.tolist()], dtype=torch.int32)
    prefix = torch.cumsum(counts, 0, dtype=torch.int32) - counts
    pts = [torch.zeros(1, 3, dtype=torch.int16)]
    per_level = [1]
    pos = 0
    bit = torch.arange(8)
    cur = torch.zeros(1, 3, dtype=torch.long)
    for _ in range(level):
        n = cur.shape[0]
        present = ((octree[pos:pos + n].long()[:, None] >> bit[None, :]) & 1).bool()
        pos += n
        child = torch.stack([(bit >> 2) & 1, (bit >> 1) & 1, bit & 1], -1)                       # child i = 4x + 2y + z
        cur = (cur[:, None, :] * 2 + child[None, :, :])[present]
        pts.append(cur.to(torch.int16))
        per_level.append(cur.shape[0])
    n_lv = torch.tensor(per_level + [0], dtype=torch.int32)
    start = torch.cumsum(n_lv, 0, dtype=torch.int32) - n_lv
    start[-1] = int(sum(per_level))
    return dict(blas_octree=octree, blas_points=torch.cat(pts), blas_prefix=prefix, blas_pyramid=torch.stack([n_lv, start]))


def save_reference_state_dict(pipeline, permuto_names=("lattice_values", "random_shift_per_level")):
    """The inverse of load_reference_state_dict(): a state dict with the key names a reference pipeline's own `state_dict()` has, so
    that `config_parser.py:757-776` (`model_format` params_only / params_only_ignore_missmatch / state_dict, all `strict=False`)
    can load a model trained here.  Decoders and `camera_extrinsics` keep their names; a hash grid's table goes out as
    `nef.<grid>.embedder.embeddings.<level>.weight` [2^log2T, F] per level (grids/hash_grid_torch.py:61-62); a permutohedral grid's as
    `nef.<grid>.embedder.<permuto_names[0]>` [L, capacity, F] and its shift as `...<permuto_names[1]>` [L, 3] - the parameter names of
    the third-party `permutohedral_encoding.PermutoEncoding` module (recalled; pass the names of the installed
    version if they differ); the occupancy as wisp's four SPC buffers (spc_buffers)."""
    from .grids import PermutoGridHIP
    nef = pipeline.nef
    out = {}
    for k, v in nef.state_dict().items():
        if "decoder" in k:
            out["nef." + k] = v.detach().float().cpu().clone()
    if hasattr(pipeline, "camera_extrinsics"):
        out["camera_extrinsics"] = pipeline.camera_extrinsics.detach().cpu().clone()
    for name in ("grid", "delta_grid"):
        grid = getattr(nef, name, None)
        if grid is None:
            continue
        prefix = "nef.%s." % name
        tab = grid.tables.detach().float().cpu()
        if isinstance(grid, PermutoGridHIP):
            out[prefix + "embedder." + permuto_names[0]] = tab.clone()
            out[prefix + "embedder." + permuto_names[1]] = grid.random_shift_per_level.detach().float().cpu().clone()
        else:
            for lv in range(tab.shape[0]):
                out[prefix + "embedder.embeddings.%d.weight" % lv] = tab[lv].clone()
        for k, v in spc_buffers(grid.blas_bits, grid.blas_level).items():
            out[prefix + k] = v
    return out
