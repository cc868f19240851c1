"""pychain_amd: MI355X-native LF-MMI loss behind the pychain API.

    from pychain_amd import ChainGraph, ChainGraphBatch, ChainFunction, ChainLoss

(`import pychain` resolves to the same objects through the alias package at the
repository root, so code written against the reference imports unchanged.)
"""
import os as _os


def _more_hardware_queues():
    """The loss runs on three streams and RCCL's collectives on a fourth; the HIP runtime maps streams onto four hardware queues by
    default, and a loss stream that shares a queue with RCCL's serialises behind it (measured: +37 % per step, bench.py).
    GPU_MAX_HW_QUEUES is read when the runtime initialises: set here - unless the user set it - while nothing has touched the
    device yet; a process that imports this package AFTER its first device call sets the variable itself."""
    if "GPU_MAX_HW_QUEUES" in _os.environ:
        return
    try:
        import torch
        if torch.cuda.is_initialized():
            return
    except Exception:
        pass
    _os.environ["GPU_MAX_HW_QUEUES"] = "16"


_more_hardware_queues()

from .graph import ChainGraph, ChainGraphBatch  # noqa: F401
from .loss import ChainFunction, ChainLoss, ChainLossFunction  # noqa: F401
