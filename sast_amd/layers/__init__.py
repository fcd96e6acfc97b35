from .sast import SAST_block, MS_WSA, PositiveLinear, get_score_index_2d21d, get_score_index_with_padding  # noqa: F401
from .ops import (window_partition, window_reverse, grid_partition, grid_reverse, LayerScale, MLP, GLU,  # noqa: F401
                  ConvDownsampling_Cf2Cl, get_downsample_layer_Cf2Cl, nChw_2_nhwC, nhwC_2_nChw, LayerNorm)
from .rnn import DWSConvLSTM2d  # noqa: F401
