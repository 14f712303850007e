"""State carried between think iterations (mirrors reference src/LanPaint/types.py:6-9)."""
from typing import NamedTuple, Optional

import torch


class LangevinState(NamedTuple):
    v: Optional[torch.Tensor]      # always None: only the overdamped scheme is live (lanpaint.py:286)
    C: Optional[torch.Tensor]
    x0: Optional[torch.Tensor]
