"""Reference-checkpoint import (config_parser.py:753-776): SPC octree <-> occupancy bitfield round trip and
state_dict mapping by name / shape.  CPU only (host logic)."""
import torch

from pagnerf_amd import checkpoint as ck


def test_octree_bitfield_round_trip():
    g = torch.Generator().manual_seed(0)
    for level in (1, 2, 3, 5):
        R = 2 ** level
        mask = torch.rand(R ** 3, generator=g) > 0.6
        mask[0] = True
        bits = ck.mask_to_bits(mask)
        octree = ck.bits_to_octree(bits, level)
        assert octree.dtype == torch.uint8 and int(octree[0]) != 0
        assert torch.equal(ck.octree_to_bits(octree, level), bits)
    # hand-made level-1 tree: children 0 (x=y=z=0) and 5 (x=1,y=0,z=1) -> cells 0 and (1*2+0)*2+1 = 5
    bits = ck.octree_to_bits(torch.tensor([0b00100001], dtype=torch.uint8), 1)
    assert int(bits[0]) == (1 << 0) | (1 << 5)


def test_state_dict_mapping_by_name_and_shape():
    import pagnerf_amd
    nef = pagnerf_amd.PanopticDeltaNeF(grid_type="PermutoGrid", num_lods=4, feature_dim=2, num_classes=3, num_instances=5,
                                       sem_num_layers=1, inst_num_layers=2, panoptic_features_type="delta", capacity_log_2=6,
                                       delta_capacity_log_2=6, blas_level=2)
    nef.grid.init_from_scales()
    nef.delta_grid.init_from_scales()
    pipe = pagnerf_amd.Pipeline(nef, pagnerf_amd.PanopticPackedRFTracer())
    g = torch.Generator().manual_seed(1)
    ref = {}
    for k, v in nef.state_dict().items():
        if "decoder" in k:
            ref["nef." + k] = torch.randn(v.shape, generator=g)
    mask = torch.rand(64, generator=g) > 0.5
    for name in ("grid", "delta_grid"):
        ref["nef.%s.embedder.lattice_values" % name] = torch.randn(4, 64, 2, generator=g)      # third-party module's own name
        ref["nef.%s.embedder.random_shift" % name] = torch.randn(4, 3, generator=g)
        ref["nef.%s.blas_octree" % name] = ck.bits_to_octree(ck.mask_to_bits(mask), 2)
        ref["nef.%s.blas_points" % name] = torch.zeros(3, 3)
    ref["something.else"] = torch.zeros(1)
    unused = ck.load_reference_state_dict(pipe, ref)
    assert unused == ["something.else"]
    assert torch.equal(nef.decoder_inst.lout.weight, ref["nef.decoder_inst.lout.weight"])
    assert torch.equal(nef.grid.tables, ref["nef.grid.embedder.lattice_values"])
    assert torch.equal(nef.delta_grid.random_shift_per_level, ref["nef.delta_grid.embedder.random_shift"])
    assert torch.equal(nef.grid.occupancy_mask(), mask) and torch.equal(nef.delta_grid.occupancy_mask(), mask)


def test_export_round_trip_with_reference_key_names(tmp_path):
    """save_reference_state_dict(): the key names of a reference pipeline's own state_dict (decoders by name, hash tables as
    embedder.embeddings.<level>.weight - grids/hash_grid_torch.py:61-62 -, wisp's four SPC buffers - grids/permuto_grid.py:33-38), and
    load_reference_state_dict() of it into a fresh pipeline restores tables, shifts, decoders and occupancy; the SPC buffers are
    consistent with each other (points per level = pyramid, child counts = prefix)."""
    import pagnerf_amd
    g = torch.Generator().manual_seed(3)
    for grid_type in ("PermutoGrid", "HashGridTorch"):
        def make(seed):
            torch.manual_seed(seed)
            kw = dict(capacity_log_2=6, delta_capacity_log_2=6) if grid_type == "PermutoGrid" else dict(codebook_bitwidth=6)
            nef = pagnerf_amd.PanopticDeltaNeF(grid_type=grid_type, num_lods=4, feature_dim=2, num_classes=3, num_instances=5, sem_num_layers=1,
                                               inst_num_layers=2, panoptic_features_type="delta", blas_level=3, **kw)
            for gr in (nef.grid, nef.delta_grid):
                if grid_type == "PermutoGrid":
                    gr.init_from_scales(random_shift=torch.randn(4, 3) * 10, tables=torch.randn(4, 64, 2))
                else:
                    gr.init_from_resolutions([16, 16, 16, 64])
                    gr.tables.data.copy_(torch.randn(4, 64, 2))
            return pagnerf_amd.Pipeline(nef, pagnerf_amd.PanopticPackedRFTracer())
        src, dst = make(1), make(2)
        mask = torch.rand(512, generator=g) > 0.7
        mask[7] = True
        for gr in (src.nef.grid, src.nef.delta_grid):
            gr.blas_init(mask)
        sd = ck.save_reference_state_dict(src)
        path = tmp_path / ("%s.pth" % grid_type)
        torch.save(sd, path)
        sd = torch.load(path)
        assert all(isinstance(v, torch.Tensor) for v in sd.values())
        assert "nef.decoder_inst.layers.1.weight" in sd and "nef.decoder_density.lout.bias" in sd
        if grid_type == "HashGridTorch":
            assert sd["nef.grid.embedder.embeddings.3.weight"].shape == (64, 2) and "nef.delta_grid.embedder.embeddings.0.weight" in sd
        else:
            assert sd["nef.grid.embedder.lattice_values"].shape == (4, 64, 2) and sd["nef.delta_grid.embedder.random_shift_per_level"].shape == (4, 3)
        oct_, pts, pre, pyr = (sd["nef.grid.blas_" + k] for k in ("octree", "points", "prefix", "pyramid"))
        assert oct_.dtype == torch.uint8 and pts.dtype == torch.int16 and pyr.shape == (2, 5)
        assert int(pyr[0, 3]) == int(mask.sum()) and int(pyr[1, 4]) == pts.shape[0] == int(pyr[0].sum()) and int(pyr[0, 0]) == 1
        assert int(pre[0]) == 0 and int(pre[-1]) + bin(int(oct_[-1])).count("1") == int(pyr[0, 1:4].sum())
        leaf = pts[int(pyr[1, 3]):int(pyr[1, 3]) + int(pyr[0, 3])].long()
        cells = torch.zeros(512, dtype=torch.bool)
        cells[(leaf[:, 0] * 8 + leaf[:, 1]) * 8 + leaf[:, 2]] = True
        assert torch.equal(cells, mask)
        assert not torch.equal(dst.nef.grid.tables, src.nef.grid.tables)
        unused = ck.load_reference_state_dict(dst, sd)
        assert unused == [], unused
        for a, b in zip(src.nef.state_dict().items(), dst.nef.state_dict().items()):
            assert a[0] == b[0] and torch.equal(a[1], b[1]), a[0]
        assert torch.equal(dst.nef.grid.occupancy_mask(), mask) and torch.equal(dst.nef.delta_grid.occupancy_mask(), mask)
