"""cmtf_pls_amd: MI355X-native NIPALS engine behind the tPLS / ctPLS API of meyer-lab/cmtf-pls.

Importing the package does not need a GPU; fitting does (there is no CPU fallback).
"""
__version__ = "0.1.0"

from .cmtf import ctPLS  # noqa: E402,F401
from .engine import EngineOptions  # noqa: E402,F401
from .tpls import tPLS  # noqa: E402,F401
