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
