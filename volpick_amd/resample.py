"""Sampling-rate conversion of the traces handed to annotate()/classify() — host-side stream preparation, as in the
reference's path: SeisBench ``WaveformModel.annotate`` resamples every trace whose rate differs from the model's before
it groups them (``seisbench_requirement >= 0.4.0``, Final_models/**/volpick.json.v1:8; un-vendored, restated here from the
published SeisBench / ObsPy algorithm — parity unpinned, like the rest of the stream handling):

    rate % model_rate == 0 :  trace.filter("lowpass", freq=model_rate / 2, zerophase=True)
                              trace.decimate(rate // model_rate, no_filter=True)
    otherwise              :  trace.resample(model_rate, no_filter=True)       (ObsPy: Fourier method, Hann window)

ObsPy's building blocks, restated with the scipy calls ObsPy itself makes:
  * ``obspy.signal.filter.lowpass``: 4-corner Butterworth as second-order sections, forward pass + time-reversed pass;
  * ``Trace.decimate(no_filter=True)``: every factor-th sample;
  * ``Trace.resample(window="hann")``: real FFT, spectrum multiplied by a Hann window centred on DC, real and imaginary
    parts linearly interpolated onto the frequency grid of the new length, inverse real FFT, amplitude rescaled.
"""
from __future__ import annotations

import warnings

import numpy as np


def lowpass_zerophase(data, freq, df, corners=4):
    from scipy.signal import iirfilter, sosfilt, zpk2sos

    fe = 0.5 * df
    f = freq / fe
    if f > 1:
        f = 1.0
        warnings.warn("Selected corner frequency is above Nyquist. Setting Nyquist as high corner.")
    z, p, k = iirfilter(corners, f, btype="lowpass", ftype="butter", output="zpk")
    sos = zpk2sos(z, p, k)
    firstpass = sosfilt(sos, data)
    return sosfilt(sos, firstpass[::-1])[::-1]


def resample_fourier(data, rate_in, rate_out, window="hann"):
    from scipy.fftpack import irfft, rfft
    from scipy.signal import get_window

    data = np.asarray(data)
    npts = len(data)
    factor = rate_in / float(rate_out)
    x = rfft(data if data.dtype.kind == "f" else data.astype(np.float64))
    x = np.insert(x, 1, x.dtype.type(0))
    if npts % 2 == 0:
        x = np.append(x, [0])
    x_r = x[::2]
    x_i = x[1::2]
    if window is not None:
        large_w = np.fft.ifftshift(get_window(window, npts))
        x_r *= large_w[: npts // 2 + 1]
        x_i *= large_w[: npts // 2 + 1]
    num = int(npts / factor)
    df = 1.0 / (npts * (1.0 / rate_in))
    d_large_f = 1.0 / num * rate_out
    f = df * np.arange(0, npts // 2 + 1, dtype=np.int32)
    n_large_f = num // 2 + 1
    large_f = d_large_f * np.arange(0, n_large_f, dtype=np.int32)
    large_y = np.zeros(2 * n_large_f)
    large_y[::2] = np.interp(large_f, f, x_r)
    large_y[1::2] = np.interp(large_f, f, x_i)
    large_y = np.delete(large_y, 1)
    if num % 2 == 0:
        large_y = np.delete(large_y, -1)
    return irfft(large_y) * (float(num) / float(npts))


def resample_array(data, rate_in, rate_out):
    """One trace's samples at ``rate_in`` -> samples at ``rate_out`` by the SeisBench rule (module docstring)."""
    if np.ma.isMaskedArray(data):
        raise NotImplementedError("masked traces cannot be resampled; split the stream at its gaps first")
    rate_in, rate_out = float(rate_in), float(rate_out)
    if rate_in == rate_out:
        return np.asarray(data)
    if rate_in % rate_out == 0:
        y = lowpass_zerophase(np.asarray(data, dtype=np.float64), rate_out * 0.5, rate_in)
        return np.ascontiguousarray(y[:: int(rate_in / rate_out)])
    return resample_fourier(data, rate_in, rate_out)


def resample_trace(tr, rate_out, copy=True):
    """A trace at ``rate_out``: the trace itself if it already is, a resampled copy (or, with ``copy=False``, the trace
    resampled in place, as upstream does) otherwise.  Works on ``volpick_amd.Trace`` and on ObsPy traces."""
    rate_in = float(tr.stats.sampling_rate)
    if abs(rate_in - rate_out) <= 1e-6:
        return tr
    out = tr.copy() if copy else tr
    out.data = resample_array(out.data, rate_in, rate_out)
    out.stats.sampling_rate = rate_out
    return out
