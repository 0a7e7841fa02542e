"""MI355X-native VITS fine-tune / inference hot path (gfx950 HIP kernels behind include/vitsmi.h).

Host-side mirror of the reference's operator surface: `models.SynthesizerTrn`,
`models.MultiPeriodDiscriminator`, `monotonic_align.maximum_path`, `commons`, `modules`,
`attentions`, `transforms`, `mel_processing`, `losses`."""
from . import _lib  # noqa: F401
from . import rng  # noqa: F401
from . import monotonic_align  # noqa: F401
from . import kernels  # noqa: F401
from . import commons  # noqa: F401
from . import transforms  # noqa: F401
from . import modules  # noqa: F401
from . import attentions  # noqa: F401
from . import mel_processing  # noqa: F401
from . import losses  # noqa: F401
from . import models  # noqa: F401
from .models import MultiPeriodDiscriminator, SynthesizerTrn  # noqa: F401
