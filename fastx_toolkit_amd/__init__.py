"""fastx_toolkit_amd -- MI355X-native engine for the fastx_toolkit hot path.

The product is the C-ABI shared library (include/fxg.h, built from csrc/) and the C host layer in
host/.  This Python package is the harness around it: build helpers, a ctypes view used by the tests
and bench.py, and the one-process-per-GPU sharding glue over torch.distributed (RCCL).
"""
from .engine import (Engine, FxgError, make_params, load_library,  # noqa: F401
                     STAGE_CLIP, STAGE_QTRIM, STAGE_QFILTER, STAGE_REVCOMP, STAGE_FTRIM, STAGE_FTRIM_END, STAGE_MASK, STAGE_ARTIFACTS, STAGE_NFILTER,
                     CLIP_DISCARD_NON_CLIPPED, CLIP_DISCARD_CLIPPED, CLIP_KEEP_N, CLIP_ADAPTER_ONLY)

__all__ = ["Engine", "FxgError", "make_params", "load_library"]
