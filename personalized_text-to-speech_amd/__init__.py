"""MI355X-native VITS fine-tune / inference hot path (gfx950 HIP kernels behind include/vitsmi.h)."""
from . import _lib  # noqa: F401
from . import monotonic_align  # noqa: F401
