"""MI355X-native nearest-neighbour engine for featureform/embeddinghub's kNN hot path.

The product is embeddinghub_amd/lib/libehx.so (hand-written HIP for gfx950 behind the C ABI of
include/ehx.h).  This package is the Python host-side mirror of the reference's Python surface
(sdk/python/offlinehub.py) plus a numpy/torch-facing wrapper used by tests and bench.py.
"""
from . import _lib
from ._lib import (DTYPE_F16, DTYPE_F32, EhxError, METRIC_COSINE, METRIC_IP, METRIC_L2SQ, MODE_FLAT, MODE_GRAPH, SCAN_AUTO, SCAN_F16, SCAN_F32, SEED_CORPUS,
                   SEED_QUERY)
from .space import Space, nearest_neighbor_rpc

__all__ = ["Space", "nearest_neighbor_rpc", "EhxError", "METRIC_L2SQ", "METRIC_IP", "METRIC_COSINE",
           "MODE_FLAT", "MODE_GRAPH", "DTYPE_F32", "DTYPE_F16", "SCAN_AUTO", "SCAN_F32", "SCAN_F16", "SEED_CORPUS", "SEED_QUERY"]
