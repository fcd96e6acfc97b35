"""Mirror of models/layers/rnn.py: ConvLSTM with a 1x1 conv over cat(x, h) (rnn.py:7-69)."""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn as nn

from .. import functional as SF


class DWSConvLSTM2d(nn.Module):
    """NCHW in / out like the reference; internally channels-last rows.  Only `dws_conv=False` (the shipped
    config, config/model/sast_yolox/default.yaml:39) is implemented; dropout on the cell update must be 0."""

    def __init__(self, dim: int, dws_conv: bool = True, dws_conv_only_hidden: bool = True, dws_conv_kernel_size: int = 3,
                 cell_update_dropout: float = 0.):
        super().__init__()
        assert isinstance(dws_conv, bool) and isinstance(dws_conv_only_hidden, bool)
        if dws_conv:
            raise NotImplementedError("sast_amd: the depth-wise 3x3 variant (dws_conv=True) is not implemented")
        if cell_update_dropout:
            raise NotImplementedError("sast_amd: cell_update_dropout > 0 is not implemented")
        self.dim = dim
        self.conv3x3_dws = nn.Identity()
        self.conv1x1 = nn.Conv2d(in_channels=dim * 2, out_channels=dim * 4, kernel_size=1)
        self.conv_only_hidden = dws_conv_only_hidden

    def forward_nhwc(self, x, h_and_c_previous=None, two_h=False):
        """two_h: (h1, h1_alias, c1) -- two handles on h1 for its two consumers (functional.conv_lstm)"""
        h0, c0 = (None, None) if h_and_c_previous is None else h_and_c_previous
        return SF.conv_lstm(x, h0, c0, self.conv1x1.weight, self.conv1x1.bias, two_h=two_h)

    def forward(self, x: torch.Tensor, h_and_c_previous: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
        """x, h, c: (N C H W) -> (h_t, c_t) (N C H W) (channels-last memory)."""
        hc = None
        if h_and_c_previous is not None:
            hc = (SF.as_nhwc(h_and_c_previous[0]), SF.as_nhwc(h_and_c_previous[1]))
        h1, c1 = self.forward_nhwc(SF.as_nhwc(x), hc)
        return SF.as_nchw_view(h1), SF.as_nchw_view(c1)
