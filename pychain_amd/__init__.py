"""pychain_amd: MI355X-native LF-MMI loss behind the pychain API."""
from .graph import ChainGraph, ChainGraphBatch  # noqa: F401
