"""Process-wide GPU context (one process per GPU; LOCAL_RANK selects the device)."""
import logging
import os

from . import _native

logging.basicConfig(level=logging.INFO, format="%(asctime)s [%(levelname)s] %(message)s",
                    datefmt="%y-%m-%d %H:%M:%S")   # same format as reference RunCmdsMP.py:13-15
logger = logging.getLogger("subphaser_amd")

_ctx = None


def get_context(device=None, stream=None):
    """The shared Context, created on first use.  Raises when the HIP library or the GPU is missing."""
    global _ctx
    if _ctx is None or _ctx.h is None:
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        _ctx = _native.Context(device, stream)
    return _ctx


def set_context(ctx):
    global _ctx
    _ctx = ctx


def close_context():
    global _ctx
    if _ctx is not None:
        _ctx.close()
        _ctx = None
