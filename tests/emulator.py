"""Numpy emulation of conv_mfma_kernel's index arithmetic (volpick_amd/csrc/conv_mfma.h),
fed with the library's own packed MFMA A-fragments and folded biases.  Test-only: lets the
CPU suite check BN folding, polyphase A-matrix construction, fragment packing and the
layer geometry against the torch oracle without a GPU."""
import ctypes as C

import numpy as np

from volpick_amd import _lib


def flat_weights(model_kind, npz):
    lib = _lib.load()
    parts = []
    for i in range(lib.vp_param_count(model_kind)):
        name = lib.vp_param_name(model_kind, i).decode()
        a = np.asarray(npz[name], dtype=np.float32).ravel()
        assert a.size == lib.vp_param_size(model_kind, i), name
        parts.append(a)
    return np.ascontiguousarray(np.concatenate(parts))


def plan_conv(model_kind, weights, index):
    """-> dict(geom, afrag, bias, name, cols, l_out) or None past the last conv layer."""
    lib = _lib.load()
    geom = (C.c_int * 13)()
    name = C.c_char_p()
    cols, l_out = C.c_int(), C.c_int()
    wp = weights.ctypes.data_as(C.c_void_p)
    n = lib.vp_debug_plan_conv(model_kind, wp, weights.size, None, index, geom, None, 0, None, 0, C.byref(name),
                               C.byref(cols), C.byref(l_out))
    if n < 0:
        return None
    g = dict(zip("cin1 cin2 cout P taps sn in_off out_off waves_m waves_n nw relu epi".split(), list(geom)))
    afrag = np.zeros(n, np.float32)
    bias = np.zeros(g["cout"] * 4, np.float32)
    lib.vp_debug_plan_conv(model_kind, wp, weights.size, None, index, geom, afrag.ctypes.data_as(C.c_void_p), n,
                           bias.ctypes.data_as(C.c_void_p), bias.size, C.byref(name), C.byref(cols), C.byref(l_out))
    return dict(geom=g, afrag=afrag, bias=bias, name=name.value.decode(), cols=cols.value, l_out=l_out.value)


def unpack_afrag(afrag, g, sets=1):
    cinp = (g["cin1"] + g["cin2"] + 3) // 4 * 4
    M = g["cout"] * g["P"]
    MT, CB, taps = M // 16, cinp // 4, g["taps"]
    per = M * cinp * taps
    out = []
    for s in range(sets):
        f = afrag[s * per:(s + 1) * per].reshape(MT, CB, taps, 4, 16)  # lane l = g*16 + n
        # A[mt*16 + n][tap][cb*4 + g]
        out.append(np.ascontiguousarray(f.transpose(0, 4, 2, 1, 3)).reshape(M, taps, cinp))
    return out


def emulate_conv(layer, src, set_index=0, sets=1):
    """src: (cin, Lin) float32 logical samples (zero outside).  Returns (cout, l_out) pre-epilogue
    conv output after bias (+ReLU), exactly as the kernel stages it."""
    g = layer["geom"]
    cin = g["cin1"] + g["cin2"]
    cinp = (cin + 3) // 4 * 4
    P, taps, sn = g["P"], g["taps"], g["sn"]
    cols, l_out = layer["cols"], layer["l_out"]
    A = unpack_afrag(layer["afrag"], g, sets)[set_index]  # (M, taps, cinp)
    Lin = src.shape[1]
    pad_l = 16
    need = sn * cols + taps + 16
    xp = np.zeros((cinp, pad_l + max(Lin, need) + 16), np.float64)
    xp[:cin, pad_l:pad_l + Lin] = src
    full = np.zeros((g["cout"] * P, cols), np.float64)
    n = np.arange(cols)
    for tap in range(taps):
        idx = pad_l + sn * n + tap + g["in_off"]
        full += A[:, tap, :].astype(np.float64) @ xp[:, idx]
    out = np.zeros((g["cout"], l_out), np.float64)
    bias = layer["bias"][set_index * g["cout"]:(set_index + 1) * g["cout"]]
    for p in range(P):
        t = P * n + p + g["out_off"]
        ok = (t >= 0) & (t < l_out)
        out[:, t[ok]] = full[p::P][:, ok]
    out += bias[:, None]
    if g["relu"]:
        out = np.maximum(out, 0)
    return out.astype(np.float32)
