"""vectordb_amd — MI355X (gfx950) implementation of Epsilla's ANN hot path behind the reference's
plugin surface.  All compute lives in lib/libepsilla_gfx950.so (HIP); see include/epsilla_gfx950.h."""
from .index import (ANNGraphSegment, Exchange, FLAT_AUTO, FLAT_MFMA, FLAT_MFMA_I8, FLAT_STREAM, GetDistFunc, GpuIndex, MODE_FLAT, MODE_GRAPH,
                    MODE_REFERENCE, VecSearchExecutor, merge_topk, merge_topk_packed, normalize_rows, traversal_gather_bytes)
from ._lib import EpsillaError, set_tuning  # noqa: F401

__all__ = ["ANNGraphSegment", "Exchange", "VecSearchExecutor", "GetDistFunc", "GpuIndex", "normalize_rows", "merge_topk", "merge_topk_packed", "traversal_gather_bytes",
           "MODE_REFERENCE", "MODE_FLAT", "MODE_GRAPH", "FLAT_AUTO", "FLAT_STREAM", "FLAT_MFMA", "FLAT_MFMA_I8", "EpsillaError", "set_tuning"]
