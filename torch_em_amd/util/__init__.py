"""Utilities of the MI355X path that sit next to the hot path (reference torch_em/util/)."""
from .prediction import predict_with_halo, predict_with_padding
