"""State carried between think iterations (mirrors reference src/LanPaint/types.py:6-9)."""
from typing import NamedTuple, Optional

import torch


class LangevinState(NamedTuple):
    v: Optional[torch.Tensor]      # always None: only the overdamped scheme is live (lanpaint.py:286)
    C: Optional[torch.Tensor]
    x0: Optional[torch.Tensor]


class FusedCFGHeads:
    """What `sampling_function_LanPaint` hands back when the two CFG combinations can be folded into
    the fused step kernel: the cond / uncond predictions of ONE batched backbone pass plus the two
    guidance scales (reference nodes.py:161-175 builds both heads eagerly:
    `uncond + (cond - uncond) * scale`, twice).  The HIP engine consumes it directly
    (LP_FL_CFG_FUSED).  For every other consumer it behaves like the reference's `(x0, x0_BIG)`
    tuple and materialises the heads on first access."""

    __slots__ = ("cond", "uncond", "scale", "scale_big", "_heads")

    def __init__(self, cond: torch.Tensor, uncond: torch.Tensor, scale: float, scale_big: float):
        self.cond, self.uncond, self.scale, self.scale_big = cond, uncond, float(scale), float(scale_big)
        self._heads = None

    def materialize(self):
        if self._heads is None:
            diff = self.cond - self.uncond
            self._heads = (self.uncond + diff * self.scale, self.uncond + diff * self.scale_big)
        return self._heads

    def __iter__(self):
        return iter(self.materialize())

    def __getitem__(self, i):
        return self.materialize()[i]

    def __len__(self):
        return 2
