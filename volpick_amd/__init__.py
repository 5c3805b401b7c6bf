"""volpick_amd — MI355X-native sliding-window phase picking with the volpick weights.

Drop-in for the ``seisbench.models`` picker API volpick users call:

    import volpick_amd as sbm
    picker = sbm.EQTransformer.from_pretrained("volpick")
    picks = picker.classify(stream, batch_size=256, overlap=5500, blinding=(500, 500),
                            stacking="avg", P_threshold=0.2, S_threshold=0.2).picks
"""
from .models import EQTransformer, PhaseNet, WaveformModel  # noqa: F401
from .picks import ClassifyOutput, Detection, DetectionList, Pick, PickList  # noqa: F401
from .stream import Stream, Trace, UTCDateTime  # noqa: F401
from ._lib import VolpickHipError  # noqa: F401
from .io import read  # noqa: F401

__version__ = "0.1.0"
