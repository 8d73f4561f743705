"""arrow_rs_amd — MI355X-native implementation of arrow-rs's columnar compute hot path
(filter / take / numeric / cmp / cast) behind the reference's function shapes.

The directory is named ``arrow-rs_amd`` (not importable as-is); ``arrow_rs_amd.py`` at
the repository root loads it under the module name ``arrow_rs_amd``.
"""
from . import _lib  # noqa: F401
from .array import *  # noqa: F401,F403
from .array import (Array, Scalar, RecordBatch, Context, DeviceBuffer, DataType, ArrowError,  # noqa: F401
                    default_context, set_default_context, pack_bits, unpack_bits, Panic, HipError)
from . import compute  # noqa: F401
from . import ffi  # noqa: F401
from . import ipc  # noqa: F401
from . import selection  # noqa: F401

__version__ = "0.1.0"
