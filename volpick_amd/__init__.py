"""volpick_amd — MI355X-native sliding-window phase picking with the volpick weights.

Drop-in for the ``seisbench.models`` picker API volpick users call:

    import volpick_amd as sbm
    picker = sbm.EQTransformer.from_pretrained("volpick")
    picks = picker.classify(stream, batch_size=256, overlap=5500, blinding=(500, 500),
                            stacking="avg", P_threshold=0.2, S_threshold=0.2).picks
"""
import os as _os

# HIP maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) when the runtime starts.  classify()
# alternates blocks over up to four device contexts, each with its own stream; with four queues for them, the null stream
# and the framework's streams, two contexts end up behind one another on one queue (EQTransformer, four contexts:
# 506 k windows/s at 4 queues, 528 k at 5-8; tools/ctx_sweep.sh).  A process that also runs torch.distributed's RCCL
# group needs eight or more for the same effect (its streams take queues too: 467 k at 6, 521-523 k at 8-16), hence 12.
# Only takes effect if no HIP call has been made yet.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")

from .models import EQTransformer, PhaseNet, WaveformModel  # noqa: F401
from .picks import ClassifyOutput, Detection, DetectionList, Pick, PickList  # noqa: F401
from .stream import Stream, Trace, UTCDateTime, pinned_array  # noqa: F401
from ._lib import VolpickHipError  # noqa: F401
from .io import read  # noqa: F401

__version__ = "0.1.0"
