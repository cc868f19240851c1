from pychain_amd.loss import ChainFunction, ChainLoss, ChainLossFunction  # noqa: F401
