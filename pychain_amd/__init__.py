"""pychain_amd: MI355X-native LF-MMI loss behind the pychain API.

    from pychain_amd import ChainGraph, ChainGraphBatch, ChainFunction, ChainLoss

(`import pychain` resolves to the same objects through the alias package at the
repository root, so code written against the reference imports unchanged.)
"""
from .graph import ChainGraph, ChainGraphBatch  # noqa: F401
from .loss import ChainFunction, ChainLoss, ChainLossFunction  # noqa: F401
