from pychain_amd.graph import ChainGraph, ChainGraphBatch  # noqa: F401
