"""pagnerf_amd - MI355X-native hot path of PAg-NeRF behind the kaolin-wisp grid / nef / tracer API.

    grid  : HashGridHIP, PermutoGridHIP           (grids.py)    <- grids/hash_grid_torch.py, grids/permuto_grid.py
    nef   : PanopticDeltaNeF, PanopticNeF         (nef.py)      <- pc_nerf/panoptic_delta_nef.py, pc_nerf/panoptic_nef.py
    tracer: PanopticPackedRFTracer                (tracer.py)   <- tracers/panoptic_packed_rf_tracer.py
    core  : Rays, RenderBuffer, Pipeline          (core.py)     <- wisp.core / wisp.models.Pipeline
    pose  : BAPipeline (learnable extrinsics)     (ba_pipeline.py) <- pc_nerf/ba_pipeline.py
    dd    : PanopticDDensityNeF / ...PackedRFTracer (dd.py)     <- pc_nerf/panoptic_dd_nef.py, tracers/panoptic_dd_packed_rf_tracer.py
    shard : ray sharding + RCCL gather/all-reduce (shard.py)
    optim : Adam (torch.optim.Adam's interface on pag_adam_step) (optim.py) <- config_parser.py:667-673, trainer.py:583

All compute goes through libpagnerf_hip.so (include/pagnerf_hip.h); there is no CPU fallback.
"""
from .core import Rays, RenderBuffer, Pipeline, batch_render       # noqa: F401
from .grids import HashGridHIP, PermutoGridHIP                     # noqa: F401
from .nef import PanopticDeltaNeF, PanopticNeF, BasicDecoder                    # noqa: F401
from .tracer import PanopticPackedRFTracer                         # noqa: F401
from .ba_pipeline import BAPipeline                                # noqa: F401
from .dd import PanopticDDensityNeF, PanopticDDensityPackedRFTracer    # noqa: F401
from . import optim                                                # noqa: F401

__version__ = "0.1.0"
