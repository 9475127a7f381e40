"""octproz_amd -- MI355X-native OCT per-A-scan processing path behind OCTproZ's plug-in boundary.

The product is liboctpipe.so (HIP kernels + C ABI, include/octpipe.h and include/octhost.h);
this package is the Python plumbing around it used by the tests and the benchmark.
"""
from . import _lib
from ._lib import OctPipeError
from .params import INTERPOLATION, OctAlgorithmParameters, WindowType, v180_benchmark_params
from .pipeline import Pipeline, PipelineGroup
from .virtual_oct import AcquisitionBuffer, Recorder, VirtualOCTSystem, synthetic_raw

__all__ = ["_lib", "OctPipeError", "INTERPOLATION", "OctAlgorithmParameters", "WindowType", "v180_benchmark_params",
           "Pipeline", "PipelineGroup", "AcquisitionBuffer", "Recorder", "VirtualOCTSystem", "synthetic_raw"]
