"""CPU oracle for the PAg-NeRF volumetric-rendering hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``pagnerf_amd/`` may import this
package; it is used by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` as the checker, never as the product.

Each module restates one piece of the reference algorithm in plain
torch-CPU / numpy and cites the reference file:line it follows
(paths relative to the upstream repository root).

Pinning status (see DESIGN.md "Oracle"):
  hash_encode      pinned   - bit-checked against tests/golden/g1_hash.npz, generated
                              by importing the reference's grids/hash_grid_torch.py
  decoders         pinned*  - against g3_nef.npz generated from the reference's
                              pc_nerf/panoptic_delta_nef.py (wisp BasicDecoder /
                              PositionalEmbedder are third-party and restated)
  render           pinned*  - against g4_tracer.npz generated from the reference's
                              tracers/panoptic_packed_rf_tracer.py (kaolin spc_render
                              ops are third-party and restated)
  lin_assign       pinned   - against g5_linassign.npz (reference loss/*.py + SciPy)
  regularizers     pinned   - against g6_reg.npz (reference loss/regularizers.py: value and
                              autograd gradient of segment_consistency_regularizer)
  permuto_encode   PARITY UNPINNED - permutohedral_encoding is an un-vendored,
                              un-versioned third-party CUDA package (README.md:45);
                              this file restates its published algorithm and is the
                              definition of record for this build.
(* = in-tree host logic pinned; third-party primitives restated from their
     public semantics.)
"""
