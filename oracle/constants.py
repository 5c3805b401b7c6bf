"""Named switches for every constant the reference repository does not pin.

Each value is the one SeisBench / ObsPy publish (see oracle/__init__.py for
why they cannot be verified offline).  The product keeps its own copies: the numerical ones are the
defaults of ``vp_default_config`` (volpick_amd/csrc/api.hip), the annotate defaults are the class-level
``_annotate_args`` of ``volpick_amd/models.py``; ``tests/test_host_logic.py::test_constants_agree_with_product``
checks both agree.
"""

# --- model numerics -------------------------------------------------------
BN_EPS = 1e-3  # BatchNorm1d eps of every BN layer in both models (TF-style)
EQT_ATTENTION_EPS = 1e-5  # additive attention: a = e / (sum(e) + eps)
EQT_LAYERNORM_EPS = 1e-14  # LayerNormalization: var + eps under the sqrt
EQT_POOL_PAD_VALUE = -1e10  # right pad before MaxPool1d(2) on odd lengths
NORM_EPS = 1e-10  # x / (peak + eps), x / (std + eps)
EQT_TAPER_SAMPLES = 6  # half-cosine taper on each window end (EQT only)
PICK_ATTENTION_WIDTH = 3  # banded attention in the P / S branches

# --- PhaseNet geometry (SeisBench PhaseNet.forward) ------------------------
PN_IN_SAMPLES = 3001
PN_KERNEL = 7
PN_STRIDE = 4
PN_DOWN_PAD = {0: (3, 3), 1: (2, 3), 2: (1, 3), 3: (2, 3)}  # (left, right) before the strided conv
PN_UP_CROP = (1, 2)  # x[:, :, 1:-2] after every ConvTranspose1d

# --- EQTransformer geometry -------------------------------------------------
EQT_IN_SAMPLES = 6000
EQT_FILTERS = (8, 16, 16, 32, 32, 64, 64)
EQT_KERNELS = (11, 9, 7, 7, 5, 5, 3)
EQT_RES_KERNELS = (3, 3, 3, 3, 2, 3, 2)
EQT_LSTM_BLOCKS = 3

# --- annotate defaults (class-level _annotate_args in SeisBench) -----------
PN_DEFAULTS = {"overlap": 1500, "blinding": (0, 0), "threshold": 0.3}
EQT_DEFAULTS = {"overlap": 1800, "blinding": (500, 500), "threshold": 0.1, "detection_threshold": 0.3}
DEFAULT_BATCH_SIZE = 256
DEFAULT_STACKING = "avg"
SAMPLING_RATE = 100.0
