"""Alias package: `import pychain` resolves to the MI355X implementation in pychain_amd
(same public names as the reference's pychain/__init__.py:1-2)."""
import sys as _sys

from pychain_amd import native as _native, simplefst as _simplefst
from pychain_amd.graph import *  # noqa: F401,F403
from pychain_amd.loss import *  # noqa: F401,F403

# `import pychain_C` / `import simplefst` written against the reference keep working
_sys.modules.setdefault("pychain_C", _native)
_sys.modules.setdefault("simplefst", _simplefst)
