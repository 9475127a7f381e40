"""The reference-side binding shown in INTEGRATION.md compiles against the reference's own headers.
Only possible where /root/reference and Qt exist (the build container); skipped elsewhere."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/octproz_project/octproz/src"
QT = "/opt/conda/include/qt"


@pytest.mark.skipif(not (os.path.isdir(REF) and os.path.isdir(QT)), reason="needs /root/reference and Qt headers")
def test_legacy_name_adapter_compiles_against_reference_headers():
    cmd = ["g++", "-std=c++11", "-fsyntax-only", "-Wall", "-DOCTPIPE_ADAPTER_NO_NOTIFIER", "-fPIC",
           "-I", os.path.join(ROOT, "include"), "-I", REF, "-I", QT, "-I", os.path.join(QT, "QtCore"),
           os.path.join(ROOT, "integration", "octproz_kernels_amd.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_adapter_exports_all_legacy_entry_points():
    src = open(os.path.join(ROOT, "integration", "octproz_kernels_amd.cpp")).read()
    for name in ["initializeCuda", "octCudaPipeline", "releaseBuffers", "destroyStreamsAndEvents", "cleanupCuda",
                 "freeCudaMem", "cuda_registerStreamingBuffers", "cuda_unregisterStreamingBuffers",
                 "cuda_registerFloatStreamingBuffers", "cuda_unregisterFloatStreamingBuffers",
                 "cuda_registerGlBufferBscan", "cuda_registerGlBufferEnFaceView", "cuda_registerGlBufferVolumeView",
                 "changeDisplayedBscanFrame", "changeDisplayedEnFaceFrame"]:
        assert ('extern "C" void %s(' % name in src) or ('extern "C" bool %s(' % name in src), name
