"""ctypes binding of libvolpick_hip.so (include/volpick_hip.h).

The library is the product: there is no Python/torch fallback for any stage of
the path.  If it is missing, :func:`load` raises with the build command.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("VOLPICK_HIP_LIB", _HERE / "libvolpick_hip.so"))

VP_MODEL_PHASENET, VP_MODEL_EQTRANSFORMER = 0, 1
VP_TRAIN_FP32, VP_TRAIN_BF16 = 0, 1
VP_NORM_PEAK, VP_NORM_STD = 0, 1
VP_STACK_AVG, VP_STACK_MAX = 0, 1
VP_MEM_HOST, VP_MEM_DEVICE = 0, 1
VP_MAX_INFLIGHT = 4
VP_ABI_VERSION = 3  # include/volpick_hip.h


class VpConfig(C.Structure):
    _fields_ = [
        ("norm", C.c_int32),
        ("norm_amp_per_comp", C.c_int32),
        ("max_batch", C.c_int32),
        ("bn_eps", C.c_float),
        ("attention_eps", C.c_float),
        ("layernorm_eps", C.c_float),
        ("norm_eps", C.c_float),
        ("taper_samples", C.c_int32),
        ("plan_flags", C.c_int32 * 8),
        ("reserved", C.c_int32 * 4),
    ]


class VpIssuedWork(C.Structure):
    _fields_ = [("mfma_f32_flop", C.c_double), ("mfma_bf16_flop", C.c_double), ("valu_flop", C.c_double)]


class VpTriggerSpec(C.Structure):
    _fields_ = [("row", C.c_int32), ("thr_on", C.c_float), ("thr_off", C.c_float)]


class VpMseedRecord(C.Structure):
    _fields_ = [
        ("offset", C.c_int64),
        ("start_us", C.c_int64),
        ("sample_rate", C.c_double),
        ("reclen", C.c_int32),
        ("data_offset", C.c_int32),
        ("nsamples", C.c_int32),
        ("encoding", C.c_int32),
        ("big_endian", C.c_int32),
        ("quality", C.c_int32),
        ("network", C.c_char * 4),
        ("station", C.c_char * 8),
        ("location", C.c_char * 4),
        ("channel", C.c_char * 4),
    ]


VP_SAMPLES_INT32, VP_SAMPLES_FLOAT32 = 0, 1


class VolpickHipError(RuntimeError):
    pass


_lib = None

# every symbol include/volpick_hip.h declares: (restype, argtypes)
_FP = C.POINTER(C.c_float)
_I64P = C.POINTER(C.c_int64)
_H = C.c_void_p
SIGNATURES = {
    "vp_default_config": (C.c_int, [C.c_int, C.POINTER(VpConfig)]),
    "vp_weight_count": (C.c_size_t, [C.c_int]),
    "vp_param_count": (C.c_int, [C.c_int]),
    "vp_param_name": (C.c_char_p, [C.c_int, C.c_int]),
    "vp_param_size": (C.c_size_t, [C.c_int, C.c_int]),
    "vp_create": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(VpConfig), C.POINTER(_H)]),
    "vp_destroy": (C.c_int, [_H]),
    "vp_in_samples": (C.c_int, [_H]),
    "vp_n_outputs": (C.c_int, [_H]),
    "vp_forward": (C.c_int, [_H, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]),
    "vp_annotate": (
        C.c_int,
        [_H, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
         _I64P, _I64P, _I64P],
    ),
    "vp_pick": (
        C.c_int,
        [_H, C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_float, _I64P, _I64P, _I64P, _FP, C.c_int,
         C.POINTER(C.c_int)],
    ),
    "vp_classify": (
        C.c_int,
        [_H, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(VpTriggerSpec),
         C.c_int, C.c_void_p, C.c_int, _I64P, _I64P, _I64P, _I64P, _I64P, _I64P, _FP, C.POINTER(C.c_int32), C.c_int,
         C.POINTER(C.c_int)],
    ),
    "vp_pick_rows": (C.c_int, [_H, C.c_void_p, C.c_int64, C.POINTER(VpTriggerSpec), C.c_int, _I64P, _I64P, _I64P, _FP,
                               C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_int)]),
    "vp_classify_submit": (
        C.c_int,
        [_H, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
         C.POINTER(VpTriggerSpec), C.c_int, C.c_void_p, C.c_int, C.c_int],
    ),
    "vp_classify_collect": (
        C.c_int,
        [_H, C.c_int, _I64P, _I64P, _I64P, _I64P, _I64P, _I64P, _FP, C.POINTER(C.c_int32), C.c_int,
         C.POINTER(C.c_int)],
    ),
    "vp_pick_windows": (
        C.c_int,
        [_H, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int,
         C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "vp_classify_multi": (
        C.c_int,
        [_H, C.c_void_p, C.c_int, _I64P, _I64P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
         C.POINTER(VpTriggerSpec), C.c_int, C.c_void_p, C.c_int, _I64P, _I64P, _I64P, _I64P, _I64P, _I64P, _FP,
         C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int, C.c_int, C.POINTER(C.c_int)],
    ),
    "vp_pick_host": (
        C.c_int,
        [C.c_void_p, C.c_int64, C.c_float, C.c_float, _I64P, _I64P, _I64P, _FP, C.c_int, C.POINTER(C.c_int)],
    ),
    "vp_window_starts": (C.c_int64, [C.c_int64, C.c_int, C.c_int, _I64P, C.c_int64]),
    "vp_last_timing": (C.c_int, [_H, _FP, _FP]),
    "vp_set_timing": (C.c_int, [_H, C.c_int]),
    "vp_stream": (C.c_void_p, [_H]),
    "vp_synchronize": (C.c_int, [_H]),
    "vp_step_count": (C.c_int, [_H]),
    "vp_step_info": (C.c_int, [_H, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_double)]),
    "vp_flops_per_window": (C.c_double, [_H]),
    "vp_step_issued_work": (C.c_int, [_H, C.c_int, C.POINTER(VpIssuedWork)]),
    "vp_step_issued_work_for_range": (C.c_int, [_H, C.c_int, C.c_int, C.c_int, C.POINTER(VpIssuedWork)]),
    "vp_step_issued_flops": (C.c_int, [_H, C.c_int, C.POINTER(C.c_double)]),
    "vp_profile_steps": (C.c_int, [_H, C.c_int, C.c_int, _FP, C.c_int]),
    "vp_profile_step_in_pipeline": (C.c_int, [_H, C.c_int, C.c_int, C.c_int, _FP]),
    "vp_profile_one_step": (C.c_int, [_H, C.c_int, C.c_int, C.c_int, _FP]),
    "vp_debug_tensor_count": (C.c_int, [_H]),
    "vp_debug_tensor_info": (C.c_int, [_H, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "vp_debug_tensor_read": (C.c_int, [_H, C.c_int, C.c_int, C.c_void_p]),
    "vp_debug_plan_conv": (
        C.c_int,
        [C.c_int, C.c_void_p, C.c_size_t, C.POINTER(VpConfig), C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_size_t,
         C.c_void_p, C.c_size_t, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.POINTER(C.c_int)],
    ),
    "vp_debug_check_halos": (C.c_int, [_H, C.c_int, _I64P, C.POINTER(C.c_char_p)]),
    "vp_rccl_available": (C.c_int, []),
    "vp_rccl_unique_id": (C.c_int, [C.c_void_p]),
    "vp_rccl_comm_init": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "vp_rccl_comm_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "vp_rccl_comm_destroy": (C.c_int, [C.c_void_p]),
    "vp_bcast_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]),
    "vp_rccl_library_path": (C.c_int, [C.c_char_p, C.c_size_t]),
    "vp_abi_version": (C.c_int, []),
    "vp_config_size": (C.c_size_t, []),
    "vp_debug_core_clock": (C.c_int, [_H, C.c_int, C.c_void_p]),
    "vp_debug_conv_clock": (C.c_int, [_H, C.c_void_p, C.c_int]),
    "vp_debug_tail_clock": (C.c_int, [_H, C.c_int, C.c_void_p]),
    "vp_mseed_scan": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(VpMseedRecord), C.c_int64, _I64P]),
    "vp_mseed_decode": (
        C.c_int,
        [C.c_int, C.c_void_p, C.c_int, C.c_size_t, C.POINTER(VpMseedRecord), _I64P, _I64P, C.c_int64, C.c_int,
         C.c_void_p, C.c_int, C.c_int64, C.c_int, C.POINTER(C.c_int32)],
    ),
    "vp_mseed_release_scratch": (C.c_int, [C.c_int, C.POINTER(C.c_size_t)]),
    "vp_mseed_decode_bench": (
        C.c_int,
        [C.c_int, C.c_void_p, C.c_size_t, C.POINTER(VpMseedRecord), _I64P, C.c_int64, C.c_int, C.c_void_p, C.c_int64,
         C.c_int, _FP],
    ),
    "vp_train_create": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(_H)]),
    "vp_train_create_dtype": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(_H)]),
    "vp_train_dtype": (C.c_int, [_H]),
    "vp_train_destroy": (C.c_int, [_H]),
    "vp_train_set_hyper": (C.c_int, [_H, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float]),
    "vp_train_set_ema": (C.c_int, [_H, C.c_float]),
    "vp_train_step": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.POINTER(C.c_double)]),
    "vp_train_synchronize": (C.c_int, [_H]),
    "vp_train_wait_inputs_consumed": (C.c_int, [_H, C.c_void_p]),
    "vp_train_inputs_consumed_upto": (C.c_longlong, [_H]),
    "vp_train_steps_enqueued": (C.c_longlong, [_H]),
    "vp_train_read": (C.c_int, [_H, C.c_int, C.c_void_p, C.c_size_t]),
    "vp_train_write_weights": (C.c_int, [_H, C.c_void_p, C.c_size_t]),
    "vp_train_predictions": (C.c_int, [_H, C.c_void_p, C.c_int]),
    "vp_train_tensor_count": (C.c_int, [_H]),
    "vp_train_tensor_info": (C.c_int, [_H, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "vp_train_tensor_read": (C.c_int, [_H, C.c_int, C.c_int, C.c_void_p]),
    "vp_train_stream": (C.c_void_p, [_H]),
    "vp_train_launch_count": (C.c_int, [_H]),
    "vp_last_error": (C.c_char_p, []),
    "vp_version": (C.c_char_p, []),
}


def load():
    """Load libvolpick_hip.so once; fail loudly if the HIP extension is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise VolpickHipError(
            f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
            f"Build it with `make -C {_HERE / 'csrc'}` or `python -c 'import __graft_entry__ as g; g.build()'`."
        )
    try:
        # PyTorch-ROCm wheels bundle their own libamdhip64.so.7 / libhsa-runtime64; a process must run on ONE
        # HIP runtime, and torch's device init fails ("No HIP GPUs are available") if the system copy was
        # loaded first.  Importing torch before the dlopen makes its copy the one both sides resolve to.
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the header and the library disagree
        fn.restype = res
        fn.argtypes = args
    if lib.vp_abi_version() != VP_ABI_VERSION or lib.vp_config_size() != C.sizeof(VpConfig):
        raise VolpickHipError(f"{LIB_PATH}: ABI {lib.vp_abi_version()} / sizeof(vp_config) {lib.vp_config_size()}, this binding "
                              f"was written for ABI {VP_ABI_VERSION} / {C.sizeof(VpConfig)}: rebuild the library")
    _lib = lib
    return lib


def check(rc: int, what: str = "volpick_hip"):
    if rc < 0:
        msg = load().vp_last_error().decode(errors="replace")
        raise VolpickHipError(f"{what} failed ({rc}): {msg}")
    return rc


def last_error() -> str:
    """The calling thread's last error message of the library."""
    return load().vp_last_error().decode(errors="replace")
