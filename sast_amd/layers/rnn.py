"""Mirror of models/layers/rnn.py: ConvLSTM with a 1x1 conv over cat(x, h), optionally behind a depth-wise k x k conv (rnn.py:7-69)."""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn as nn

from .. import functional as SF


class DWSConvLSTM2d(nn.Module):
    """NCHW in / out like the reference; internally channels-last rows.  `dws_conv=False` is the shipped config
    (config/model/sast_yolox/default.yaml:39); `dws_conv=True` (the reference class default) runs the depth-wise conv as its own
    HIP kernel (sast_dwconv_*) in front of the fused 1x1-conv + gates launch: on the previous hidden state
    (`dws_conv_only_hidden=True`, rnn.py:52-53) or on x and h with the two halves of the depth-wise weight (rnn.py:55-56: a depth-wise
    conv of cat(x, h) is the halves convolved on their own).  `cell_update_dropout > 0`: the keep mask is drawn on the device and
    applied in the gates epilogue (`SastLstmArgs.drop`)."""

    def __init__(self, dim: int, dws_conv: bool = True, dws_conv_only_hidden: bool = True, dws_conv_kernel_size: int = 3,
                 cell_update_dropout: float = 0.):
        super().__init__()
        assert isinstance(dws_conv, bool) and isinstance(dws_conv_only_hidden, bool)
        if not 0.0 <= cell_update_dropout < 1.0:
            raise ValueError(f"cell_update_dropout must be in [0, 1), got {cell_update_dropout}")
        self.cell_update_dropout = nn.Dropout(p=cell_update_dropout)      # rnn.py:34 (no parameters: the state_dict is unchanged)
        self.drop_mask_override = None     # tests: a fixed keep mask / (1 - p), NHWC, used instead of a fresh draw
        if dws_conv and (dws_conv_kernel_size % 2 == 0 or dws_conv_kernel_size > 7):
            raise NotImplementedError("sast_amd: the depth-wise conv kernel is built for odd kernel sizes up to 7")
        self.dim = dim
        dws_dim = dim if dws_conv_only_hidden else 2 * dim
        self.conv3x3_dws = nn.Conv2d(in_channels=dws_dim, out_channels=dws_dim, kernel_size=dws_conv_kernel_size,
                                     padding=dws_conv_kernel_size // 2, groups=dws_dim) if dws_conv else nn.Identity()
        self.dws_conv = dws_conv
        self.conv1x1 = nn.Conv2d(in_channels=dim * 2, out_channels=dim * 4, kernel_size=1)
        self.conv_only_hidden = dws_conv_only_hidden

    def forward_nhwc(self, x, h_and_c_previous=None, two_h=False):
        """two_h: (h1, h1_alias, c1) -- two handles on h1 for its two consumers (functional.conv_lstm)"""
        h0, c0 = (None, None) if h_and_c_previous is None else h_and_c_previous
        if self.dws_conv:
            dw, db, C = self.conv3x3_dws.weight, self.conv3x3_dws.bias, self.dim
            if h0 is None:                       # the reference convolves the zero state: the result is the bias, not zero
                h0 = torch.zeros_like(x)
            if self.conv_only_hidden:
                h0 = SF.dwconv(h0, dw, db)
            else:
                x = SF.dwconv(x, dw, db, 0)          # channels [0, C) of the depth-wise parameters: the x half of cat(x, h)
                h0 = SF.dwconv(h0, dw, db, C)        # channels [C, 2C): the h half
        return SF.conv_lstm(x, h0, c0, self.conv1x1.weight, self.conv1x1.bias, two_h=two_h, drop_mask=self._cell_dropout_mask(x))

    def _cell_dropout_mask(self, x):
        """rnn.py:64: `cell_input = self.cell_update_dropout(th.tanh(cell_input))`.  nn.Dropout in training mode multiplies by a
        Bernoulli(1 - p) keep mask divided by (1 - p); the mask is drawn here with torch's generator of x's device (as the reference's
        would be on a GPU) and applied inside the fused gates epilogue and its backward."""
        p = self.cell_update_dropout.p
        if not self.training or p == 0.0:
            return None
        if self.drop_mask_override is not None:
            return self.drop_mask_override
        return torch.empty_like(x).bernoulli_(1.0 - p).div_(1.0 - p)

    def forward(self, x: torch.Tensor, h_and_c_previous: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
        """x, h, c: (N C H W) -> (h_t, c_t) (N C H W) (channels-last memory)."""
        hc = None
        if h_and_c_previous is not None:
            hc = (SF.as_nhwc(h_and_c_previous[0]), SF.as_nhwc(h_and_c_previous[1]))
        h1, c1 = self.forward_nhwc(SF.as_nhwc(x), hc)
        return SF.as_nchw_view(h1), SF.as_nchw_view(c1)
