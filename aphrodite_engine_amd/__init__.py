"""aphrodite-engine_amd: MI355X (gfx950) native quantized-inference hot path for
PygmalionAI/aphrodite-engine -- paged attention + GPTQ/AWQ/FP8 linears behind
the reference's own plugin surface (``_custom_ops`` / ``QuantizeMethodBase`` /
``AttentionBackend``).  See DESIGN.md and INTEGRATION.md."""
__version__ = "0.1.0"

from . import _lib  # noqa: F401


def library_path() -> str:
    return _lib.LIB_PATH


def load_library():
    """Load libaphrodite_mi355x.so; raises ImportError if it was not built."""
    return _lib.lib()
