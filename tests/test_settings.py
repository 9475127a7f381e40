"""Settings INI and curve CSV readers: the reference's own published settings file
must reproduce the benchmark parameter set.  CPU only."""
import os

import numpy as np

from octproz_amd import params as P

# the [processing] / [streaming] / [Virtual OCT System] groups of the reference's
# performance/v180/20250504_performance_v180_gtx1080/20250504_octproz_settings.ini (values only: a fixture)
V180_INI = """[General]
timestamp=20250504_141131540

[record]
record_processed=true
save_as_32_bit_float=false

[processing]
addend=0
bitshift=false
coeff=1
dispersion_compensation=true
dispersion_compensation_d0=0
dispersion_compensation_d1=97
dispersion_compensation_d2=-96.625
dispersion_compensation_d3=-0.375
fixed_pattern_removal=true
fixed_pattern_removal_continuously=false
fixed_pattern_removal_bscans=1
flip_bscans=false
log=true
max=100
min=-30
resampling=true
resampling_c0=0.535239
resampling_c1=871.817574
resampling_c2=-170.633784
resampling_c3=97.249716
resampling_interpolation=1
sinusoidal_scan_correction=false
window_center_position=0.5
window_fill_factor=0.95
window_type=0
windowing=true
background_removal=false
background_removal_window_size=8
custom_resampling=false
custom_resampling_filepath=
post_processing_background_removal=false
post_processing_background_removal_offset=0
post_processing_background_removal_weight=1

[streaming]
streaming_enabled=true
streaming_skip=0

[Virtual%20OCT%20System]
bit_depth=12
buffers_from_file=2
buffers_per_volume=1
depth=256
file_path=C:/test_data_raw.raw
height=512
wait_time=0
width=1024
copy_file_to_ram=true
bscan_offset=0
sync_with_processing=true
"""


def test_reference_settings_file_reproduces_the_benchmark_parameters(tmp_path):
    path = os.path.join(tmp_path, "settings.ini")
    open(path, "w").write(V180_INI)
    p, vsys = P.load_settings_ini(path)
    want = P.v180_benchmark_params(1024, 512, 256)
    for name, _ in P.PipeParams._fields_:
        if name in ("streamToHost",):
            continue
        assert getattr(p, name) == getattr(want, name), name
    assert p.streamToHost == 1  # the published run had streaming enabled
    assert (p.samplesPerLine, p.ascansPerBscan, p.bscansPerBuffer, p.buffersPerVolume, p.bitDepth) == (1024, 512, 256, 1, 12)
    for a, b in ((p.resampleCurve, want.resampleCurve), (p.dispersionCurve, want.dispersionCurve), (p.windowCurve, want.windowCurve)):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert vsys["file_path"] == "C:/test_data_raw.raw" and vsys["buffers_from_file"] == 2 and vsys["sync_with_processing"]


def test_curve_csv_round_trip_and_reference_format(tmp_path):
    path = os.path.join(tmp_path, "resampling.csv")
    curve = np.linspace(0.25, 1020.5, 1024, dtype=np.float32)
    P.save_curve_csv(path, curve)
    lines = open(path).read().splitlines()
    assert lines[0] == "Sample Number;Sample Value" and lines[1].startswith("0;") and len(lines) == 1025
    back = P.load_curve_csv(path)
    # six significant digits survive: the writer is QTextStream << float (octalgorithmparametersmanager.cpp:32-45), byte for byte
    # what the reference writes (tests/test_host_reference.py) -- not a lossless format
    assert np.allclose(back, curve, rtol=5e-6, atol=0) and lines[2] == "1;1.24731" and lines[-1] == "1023;1020.5"
    # a file as the reference writes it (QTextStream default float formatting, extra column, empty field)
    open(path, "w").write("Sample Number;Sample Value\n0;1.5\n1;2.25;x\n2;\n3;-7e-1\n")
    assert np.allclose(P.load_curve_csv(path), [1.5, 2.25, 0.0, -0.7])


def test_custom_resampling_curve_from_settings(tmp_path):
    csv = os.path.join(tmp_path, "resampling.csv")
    P.save_curve_csv(csv, np.arange(1024, dtype=np.float32) * 0.5)
    ini = os.path.join(tmp_path, "s.ini")
    open(ini, "w").write(V180_INI.replace("custom_resampling=false", "custom_resampling=true").replace("custom_resampling_filepath=", "custom_resampling_filepath=" + csv))
    p, _ = P.load_settings_ini(ini)
    assert p.useCustomResampleCurve and np.array_equal(p.resampleCurve, np.arange(1024, dtype=np.float32) * 0.5)


def test_settings_ini_round_trip(tmp_path):
    """octhost_save_settings_ini writes what octhost_load_settings_ini (and OCTproZ) reads: every processing field, the curve
    coefficients and the virtual-system group survive a round trip, booleans as true / false, the group name %20-escaped"""
    from octproz_amd.params import load_settings_ini, save_settings_ini, v180_benchmark_params
    p = v180_benchmark_params(2048, 300, 7, buffers_per_volume=3)
    p.bitshift, p.bscanFlip, p.backgroundRemoval, p.rollingAverageWindowSize = 1, 1, 1, 33
    p.postProcessBackgroundRemoval, p.postProcessBackgroundWeight, p.postProcessBackgroundOffset = 1, 0.75, -0.125
    p.streamToHost, p.streamingBuffersToSkip, p.sinusoidalScanCorrection = 1, 4, 1
    p.signalMultiplicator, p.signalAddend = 2.5, -0.25
    path = str(tmp_path / "settings.ini")
    vs = {"file_path": "/data/rec.raw", "buffers_from_file": 5, "bscan_offset": 3, "wait_time_us": 11, "copy_file_to_ram": False, "sync_with_processing": True}
    save_settings_ini(path, p, vs, timestamp="20250504_141131540")
    text = open(path).read()
    assert "[Virtual%20OCT%20System]" in text and "bitshift=true" in text and "timestamp=20250504_141131540" in text
    q, vsys = load_settings_ini(path)
    for name in ("bitshift", "bscanFlip", "signalLogScaling", "sinusoidalScanCorrection", "signalGrayscaleMin", "signalGrayscaleMax", "signalMultiplicator",
                 "signalAddend", "backgroundRemoval", "rollingAverageWindowSize", "resampling", "resamplingInterpolation", "dispersionCompensation", "windowing",
                 "fixedPatternNoiseRemoval", "continuousFixedPatternNoiseDetermination", "bscansForNoiseDetermination", "postProcessBackgroundRemoval",
                 "postProcessBackgroundWeight", "postProcessBackgroundOffset", "streamToHost", "streamingBuffersToSkip", "samplesPerLine", "ascansPerBscan",
                 "bscansPerBuffer", "buffersPerVolume", "bitDepth"):
        assert getattr(q, name) == getattr(p, name), name
    for name in ("c0", "c1", "c2", "c3", "d0", "d1", "d2", "d3", "windowCenter", "windowFillFactor"):
        assert np.float32(getattr(q, name)) == np.float32(getattr(p, name)), name
    assert int(q.window) == int(p.window)
    assert np.array_equal(q.resampleCurve, p.resampleCurve) and np.array_equal(q.windowCurve, p.windowCurve) and np.array_equal(q.dispersionCurve, p.dispersionCurve)
    assert vsys["file_path"] == "/data/rec.raw" and vsys["buffers_from_file"] == 5 and vsys["bscan_offset"] == 3 and vsys["wait_time_us"] == 11
    assert vsys["copy_file_to_ram"] is False and vsys["sync_with_processing"] is True
