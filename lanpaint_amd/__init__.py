"""lanpaint_amd -- LanPaint's Langevin "think" loop as hand-written HIP kernels for
MI355X (gfx950), behind the reference's own Python API.

    from lanpaint_amd import LanPaint            # engine, same signature as the reference
    from lanpaint_amd.nodes import KSamplerX0Inpaint, reshape_mask, NODE_CLASS_MAPPINGS

Importing the engine loads liblanpaint_hip.so and fails loudly when it is missing.
"""
from .types import FusedCFGHeads, LangevinState  # noqa: F401
from .lanpaint import LanPaint, pack_mask  # noqa: F401

__all__ = ["LanPaint", "LangevinState", "FusedCFGHeads", "pack_mask"]
__version__ = "0.1.0"
