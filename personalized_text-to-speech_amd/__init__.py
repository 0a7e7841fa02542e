"""MI355X-native VITS fine-tune / inference hot path (gfx950 HIP kernels behind include/vitsmi.h).

Host-side mirror of the reference's operator surface: `models.SynthesizerTrn`,
`models.MultiPeriodDiscriminator`, `monotonic_align.maximum_path`, `commons`, `modules`,
`attentions`, `transforms`, `mel_processing`, `losses`."""
import os as _os

# hipGraph replay on ROCm 7.x: with the runtime's AQL-packet capture enabled, device memset nodes (torch's multi-block
# reductions zero their semaphores with one; MIOpen does the same for split-k outputs) are not re-executed correctly from
# the second replay on — reductions then return stale memory (tools/dbg_disc_graph.py reproduces it).  The
# runtime reads this switch when it initialises, i.e. at the first HIP call of the process, so it is set at import.
_os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

from . import _lib  # noqa: F401,E402
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
from . import utils  # noqa: F401
from . import optim  # noqa: F401
from .models import MultiPeriodDiscriminator, SynthesizerTrn  # noqa: F401
