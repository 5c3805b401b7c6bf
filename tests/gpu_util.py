"""Helpers for the GPU parity tests: read intermediate activations out of a vp handle."""
import ctypes as C

import numpy as np

from volpick_amd import _lib


def debug_tensors(model, B):
    """{name: (B*sets, C, L) ndarray} of every activation tensor after the last forward."""
    lib = _lib.load()
    h = model._handle
    out = {}
    for i in range(lib.vp_debug_tensor_count(h)):
        name, c, l = C.c_char_p(), C.c_int(), C.c_int()
        _lib.check(lib.vp_debug_tensor_info(h, i, C.byref(name), C.byref(c), C.byref(l)))
        a = np.empty((B, c.value, l.value), np.float32)
        rc = lib.vp_debug_tensor_read(h, i, B, a.ctypes.data_as(C.c_void_p))
        if rc == -4:  # VP_ERR_UNSUPPORTED: the plan keeps this tensor in LDS (fused kernels)
            continue
        _lib.check(rc)
        out[name.value.decode()] = a
    return out


def step_profile(model, B, iters=20):
    lib = _lib.load()
    h = model._handle
    n = lib.vp_step_count(h)
    ms = (C.c_float * n)()
    _lib.check(lib.vp_profile_steps(h, B, iters, ms, n))
    rows = []
    for i in range(n):
        name, fl = C.c_char_p(), C.c_double()
        lib.vp_step_info(h, i, C.byref(name), C.byref(fl))
        rows.append((name.value.decode(), fl.value, ms[i]))
    return rows
