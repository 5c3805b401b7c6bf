"""CPU: the arithmetic behind bench.py's `roofline` object (no GPU, no library call)."""
import importlib.util
from pathlib import Path

spec = importlib.util.spec_from_file_location("bench", Path(__file__).resolve().parents[1] / "bench.py")
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def test_pipe_time_adds_the_pipes_and_never_exceeds_a_real_duration():
    # eqt_tail3_kernel: 611.9424 MFLOP per window as bf16 MFMAs -> 256 windows at 2500 TFLOP/s
    t = bench.pipe_time_s({"mfma_f32": 0.0, "mfma_bf16": 611_942_400.0, "valu": 0.0}, 256)
    assert abs(t - 256 * 611_942_400.0 / 2.5e15) < 1e-12 and 62e-6 < t < 63e-6
    # pn_window_kernel: the three pipes share the SIMD's issue, their times add
    mix = {"mfma_f32": 20_762_624.0, "mfma_bf16": 70_385_664.0, "valu": 9_219_072.0}
    t = bench.pipe_time_s(mix, 256)
    want = 256 * ((20_762_624.0 + 9_219_072.0) / 157.3e12 + 70_385_664.0 / 2.5e15)
    assert abs(t - want) < 1e-12
    assert t / 97e-6 < 1.0  # frac of a measured 97 us launch: below one by construction of the peaks
    assert bench.PEAK_FP32_TFLOPS == 157.3 and bench.PEAK_BF16_TFLOPS == 2500.0 and bench.PEAK_HBM_GBS == 8000.0
