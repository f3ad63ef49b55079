"""Utilities of the MI355X path that sit next to the hot path (reference torch_em/util/)."""
from .prediction import predict_with_halo, predict_with_halo_pipelined, predict_with_padding
from .util import (ensure_array, ensure_spatial_array, ensure_tensor, ensure_tensor_with_channels,
                   get_constructor_arguments, get_trainer, load_model, model_is_equal)
