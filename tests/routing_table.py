"""The routing table: which implementation (OCTPIPE_PATH_* bits of octpipe_debug_last_path) a configuration runs on.  Part of the
contract: every route gives the oracle's image, so a configuration that silently fell back to a slower route would pass
every parity test.  Held twice: against the pure routing function without a device (tests/test_route.py, octpipe_debug_route =
csrc/route.h) and against what the device actually ran (tests/test_gpu_api_contracts.py::test_routing_table)."""
from octproz_amd import _lib as _P

ROUTING = [
    # N, settings, sample format, route flags, expected OCTPIPE_PATH_* bits of the image launch
    (1024, {}, 0, 0, 0),
    (1024, {"dispersionCompensation": 0}, 0, 0, _P.PATH_REAL_INPUT),
    (1024, {"backgroundRemoval": 1, "rollingAverageWindowSize": 64}, 0, 0, _P.PATH_ROLL_IN_KERNEL),
    (1024, {"backgroundRemoval": 1, "rollingAverageWindowSize": 300}, 0, 0, _P.PATH_PREPARED_ROWS),
    (1024, {"backgroundRemoval": 1, "rollingAverageWindowSize": 200, "bitDepth": 16}, 0, 0, _P.PATH_PREPARED_ROWS),  # sums not exact
    (1024, {"postProcessBackgroundRemoval": 1}, 0, 0, _P.PATH_FUSED_BG),
    # sinusoidal scan correction: inside the image store of the general kernel (round 6), and the removal that follows it with it
    (1024, {"sinusoidalScanCorrection": 1}, 0, 0, _P.PATH_FUSED_SINUS),
    (1024, {"postProcessBackgroundRemoval": 1, "sinusoidalScanCorrection": 1}, 0, 0, _P.PATH_FUSED_SINUS | _P.PATH_FUSED_BG),
    (1024, {"postProcessBackgroundRemoval": 1, "sinusoidalScanCorrection": 1}, 0, _P.ROUTE_NO_FUSED_SINUS, 0),               # both in the post pass
    (1024, {"postProcessBackgroundRemoval": 1, "sinusoidalScanCorrection": 1}, 0, _P.ROUTE_NO_FUSED_BG, _P.PATH_FUSED_SINUS),  # the removal alone in the post pass
    (1024, {"sinusoidalScanCorrection": 1, "dispersionCompensation": 0}, 0, 0, _P.PATH_FUSED_SINUS),                          # outranks the real-input kernel + post pass
    (1024, {"sinusoidalScanCorrection": 1, "dispersionCompensation": 0}, 0, _P.ROUTE_NO_FUSED_SINUS, _P.PATH_REAL_INPUT),
    (1024, {"sinusoidalScanCorrection": 1, "bscanFlip": 1, "backgroundRemoval": 1, "rollingAverageWindowSize": 64}, 0, 0, _P.PATH_FUSED_SINUS | _P.PATH_ROLL_IN_KERNEL),
    (1024, {"sinusoidalScanCorrection": 1, "resamplingInterpolation": 2}, 0, 0, 0),                                           # Lanczos: post pass
    (1024, {"sinusoidalScanCorrection": 1}, 1, 0, 0),                                                                          # packed 12 bit rows: post pass
    (1024, {"sinusoidalScanCorrection": 1, "backgroundRemoval": 1, "rollingAverageWindowSize": 300}, 0, 0, _P.PATH_PREPARED_ROWS),
    (2048, {"sinusoidalScanCorrection": 1}, 0, 0, _P.PATH_FUSED_SINUS),
    (2048, {"sinusoidalScanCorrection": 1, "backgroundRemoval": 1, "rollingAverageWindowSize": 64}, 0, 0, _P.PATH_ROLL_IN_KERNEL),  # cubic + rolling at 2048: registers
    (512, {"sinusoidalScanCorrection": 1, "resamplingInterpolation": 0}, 0, 0, _P.PATH_FUSED_SINUS),
    (256, {"sinusoidalScanCorrection": 1, "resampling": 0}, 0, 0, _P.PATH_FUSED_SINUS),
    (4096, {"sinusoidalScanCorrection": 1}, 0, 0, _P.PATH_TEAM | _P.PATH_FUSED_SINUS),                                        # inside the team kernel's store too (round 6)
    (4096, {"sinusoidalScanCorrection": 1, "dispersionCompensation": 0}, 0, 0, _P.PATH_TEAM | _P.PATH_FUSED_SINUS),           # outranks the real-input team kernel + post pass
    (4096, {"sinusoidalScanCorrection": 1, "postProcessBackgroundRemoval": 1, "resamplingInterpolation": 0, "backgroundRemoval": 1, "rollingAverageWindowSize": 64}, 0, 0,
     _P.PATH_TEAM | _P.PATH_FUSED_SINUS | _P.PATH_FUSED_BG | _P.PATH_ROLL_IN_KERNEL),
    (4096, {"sinusoidalScanCorrection": 1, "backgroundRemoval": 1, "rollingAverageWindowSize": 64}, 0, 0, _P.PATH_TEAM | _P.PATH_ROLL_IN_KERNEL),  # cubic + rolling average: registers -> post pass
    (4096, {"sinusoidalScanCorrection": 1}, 0, _P.ROUTE_NO_FUSED_SINUS, _P.PATH_TEAM),
    (4096, {"sinusoidalScanCorrection": 1, "resamplingInterpolation": 2}, 0, 0, _P.PATH_TEAM),                                # Lanczos: post pass
    (8192, {"sinusoidalScanCorrection": 1}, 0, 0, _P.PATH_TEAM | _P.PATH_FUSED_SINUS),
    (8192, {"sinusoidalScanCorrection": 1, "postProcessBackgroundRemoval": 1}, 0, 0, _P.PATH_TEAM | _P.PATH_FUSED_SINUS),     # the removal as the post pass there (LDS)
    (1664, {"sinusoidalScanCorrection": 1}, 0, 0, _P.PATH_TEAM | _P.PATH_FUSED_SINUS),                                        # the reference recording's length: the two-wave team kernel (round 6)
    (1664, {"sinusoidalScanCorrection": 1, "resamplingInterpolation": 0}, 0, 0, _P.PATH_TEAM | _P.PATH_FUSED_SINUS),          # ... for every resampling mode it has
    (1664, {"sinusoidalScanCorrection": 1, "dispersionCompensation": 0, "resampling": 0}, 0, 0, _P.PATH_TEAM | _P.PATH_FUSED_SINUS),
    (1664, {"sinusoidalScanCorrection": 1, "postProcessBackgroundRemoval": 1, "backgroundRemoval": 1, "rollingAverageWindowSize": 64}, 0, 0,
     _P.PATH_TEAM | _P.PATH_FUSED_SINUS | _P.PATH_FUSED_BG | _P.PATH_ROLL_IN_KERNEL),
    (1664, {"sinusoidalScanCorrection": 1, "resamplingInterpolation": 0}, 0, _P.ROUTE_NO_FUSED_SINUS, _P.PATH_MIXED_RADIX),
    (1664, {"sinusoidalScanCorrection": 1, "resamplingInterpolation": 0}, 0, _P.ROUTE_NO_TEAM, _P.PATH_MIXED_RADIX),
    (1664, {"resamplingInterpolation": 0}, 0, _P.ROUTE_TEAM1664_ALWAYS | _P.ROUTE_NO_REAL_INPUT, _P.PATH_TEAM),
    (1664, {"sinusoidalScanCorrection": 1, "resamplingInterpolation": 2}, 0, 0, _P.PATH_MIXED_RADIX),                         # Lanczos: post pass
    (1664, {"sinusoidalScanCorrection": 1}, 1, 0, _P.PATH_TEAM | _P.PATH_PREPARED_ROWS),                                      # packed 12 bit rows: prepared, post pass
    # ... and of the kernels compiled at run time, where a lane's bins of the previous row fit in registers (round 6)
    (1000, {"sinusoidalScanCorrection": 1}, 0, 0, _P.PATH_MIXED_RADIX | _P.PATH_STATIC_PLAN | _P.PATH_FUSED_SINUS),
    (1000, {"sinusoidalScanCorrection": 1, "dispersionCompensation": 0}, 0, 0, _P.PATH_MIXED_RADIX | _P.PATH_STATIC_PLAN | _P.PATH_FUSED_SINUS),  # one A-scan per transform then
    (1000, {"sinusoidalScanCorrection": 1, "postProcessBackgroundRemoval": 1, "backgroundRemoval": 1, "rollingAverageWindowSize": 64}, 0, 0,
     _P.PATH_MIXED_RADIX | _P.PATH_STATIC_PLAN | _P.PATH_FUSED_SINUS | _P.PATH_FUSED_BG | _P.PATH_ROLL_IN_KERNEL),
    (1000, {"sinusoidalScanCorrection": 1}, 0, _P.ROUTE_NO_FUSED_SINUS, _P.PATH_MIXED_RADIX | _P.PATH_STATIC_PLAN),
    (1000, {"sinusoidalScanCorrection": 1, "resamplingInterpolation": 2}, 0, 0, _P.PATH_MIXED_RADIX | _P.PATH_STATIC_PLAN),           # Lanczos: post pass
    (1000, {"sinusoidalScanCorrection": 1}, 0, _P.ROUTE_NO_MIXEDN_STATIC, _P.PATH_MIXED_RADIX),                                          # the run-time-plan kernel: post pass
    (2000, {"sinusoidalScanCorrection": 1}, 0, 0, _P.PATH_MIXED_RADIX | _P.PATH_STATIC_PLAN | _P.PATH_FUSED_SINUS),                   # 20 bins per lane
    (2000, {"sinusoidalScanCorrection": 1, "backgroundRemoval": 1, "rollingAverageWindowSize": 64}, 0, 0, _P.PATH_MIXED_RADIX | _P.PATH_STATIC_PLAN | _P.PATH_ROLL_IN_KERNEL),  # ... too many next to the rolling average
    (3000, {"sinusoidalScanCorrection": 1}, 0, 0, _P.PATH_MIXED_RADIX | _P.PATH_STATIC_PLAN),                                          # 25 bins per lane: post pass
    (1800, {"sinusoidalScanCorrection": 1}, 0, 0, _P.PATH_MIXED_RADIX | _P.PATH_STATIC_PLAN),                                          # 30 values + 16 bins at the 168-register budget: post pass
    (2500, {"sinusoidalScanCorrection": 1}, 0, 0, _P.PATH_MIXED_RADIX | _P.PATH_STATIC_PLAN),                                          # four passes: post pass (measured faster)
    (6144, {"sinusoidalScanCorrection": 1}, 0, 0, _P.PATH_MIXED_RADIX | _P.PATH_STATIC_PLAN),                                          # two waves per A-scan, 24 bins per lane: post pass
    (1024, {"postProcessBackgroundRemoval": 1, "backgroundRemoval": 1, "rollingAverageWindowSize": 64}, 0, 0, _P.PATH_FUSED_BG | _P.PATH_ROLL_IN_KERNEL),
    (4096, {"postProcessBackgroundRemoval": 1, "backgroundRemoval": 1, "rollingAverageWindowSize": 64}, 0, 0, _P.PATH_TEAM | _P.PATH_FUSED_BG | _P.PATH_ROLL_IN_KERNEL),
    (1664, {"postProcessBackgroundRemoval": 1, "backgroundRemoval": 1, "rollingAverageWindowSize": 64}, 0, 0, _P.PATH_TEAM | _P.PATH_FUSED_BG | _P.PATH_ROLL_IN_KERNEL),
    (1024, {"postProcessBackgroundRemoval": 1}, 0, _P.ROUTE_NO_FUSED_BG, 0),
    (1024, {"postProcessBackgroundRemoval": 1}, 1, 0, _P.PATH_FUSED_BG),                      # packed 12 bit, decoded in the kernel
    (1024, {"backgroundRemoval": 1, "rollingAverageWindowSize": 8}, 1, 0, _P.PATH_PREPARED_ROWS),
    (1024, {"bitDepth": 32, "postProcessBackgroundRemoval": 1}, 5, 0, _P.PATH_PREPARED_ROWS | _P.PATH_FUSED_BG),
    (2048, {"dispersionCompensation": 0}, 0, 0, _P.PATH_REAL_INPUT),
    (4096, {}, 0, 0, _P.PATH_TEAM),
    (4096, {"dispersionCompensation": 0}, 0, 0, _P.PATH_TEAM | _P.PATH_REAL_INPUT),
    (4096, {"dispersionCompensation": 0}, 0, _P.ROUTE_NO_REAL_INPUT, _P.PATH_TEAM),
    (4096, {"backgroundRemoval": 1, "rollingAverageWindowSize": 64}, 0, 0, _P.PATH_TEAM | _P.PATH_ROLL_IN_KERNEL),
    (4096, {"backgroundRemoval": 1, "rollingAverageWindowSize": 64}, 0, _P.ROUTE_NO_TEAM, _P.PATH_ROLL_IN_KERNEL),
    (4096, {"postProcessBackgroundRemoval": 1}, 2, 0, _P.PATH_TEAM | _P.PATH_FUSED_BG),       # signed packed 12 bit
    (4096, {"resamplingInterpolation": 2}, 0, 0, _P.PATH_TEAM),                                # Lanczos on the team kernel
    (4096, {"resamplingInterpolation": 2}, 0, _P.ROUTE_NO_TEAM, 0),
    (4096, {"resamplingInterpolation": 2, "backgroundRemoval": 1, "rollingAverageWindowSize": 16}, 0, 0, _P.PATH_TEAM | _P.PATH_PREPARED_ROWS),
    (8192, {}, 0, 0, _P.PATH_TEAM),
    (8192, {"dispersionCompensation": 0}, 0, 0, _P.PATH_TEAM | _P.PATH_REAL_INPUT),
    (8192, {"backgroundRemoval": 1, "rollingAverageWindowSize": 64}, 0, 0, _P.PATH_TEAM | _P.PATH_ROLL_IN_KERNEL),
    (8192, {"bitDepth": 16}, 4, 0, _P.PATH_TEAM | _P.PATH_PREPARED_ROWS),                      # int16 comes prepared at this length
    (8192, {"resamplingInterpolation": 2}, 0, 0, _P.PATH_TEAM),
    (8192, {"resamplingInterpolation": 2}, 0, _P.ROUTE_NO_TEAM, _P.PATH_MIXED_RADIX | _P.PATH_STATIC_PLAN),   # round 6: the compiled kernel on a team of two waves takes what the dedicated team kernel leaves
    (8192, {"resamplingInterpolation": 2}, 0, _P.ROUTE_NO_TEAM | _P.ROUTE_NO_MIXEDN, _P.PATH_LIBRARY_FFT | _P.PATH_PREPARED_ROWS),
    (8192, {"resamplingInterpolation": 2, "dispersionCompensation": 0}, 0, 0, _P.PATH_TEAM),
    (1664, {}, 0, 0, _P.PATH_TEAM),
    (1664, {"resamplingInterpolation": 0}, 0, 0, _P.PATH_MIXED_RADIX),
    (1664, {"dispersionCompensation": 0}, 0, 0, _P.PATH_MIXED_RADIX | _P.PATH_REAL_INPUT),
    (1664, {"postProcessBackgroundRemoval": 1}, 0, 0, _P.PATH_TEAM | _P.PATH_FUSED_BG),
    (1664, {"postProcessBackgroundRemoval": 1, "resamplingInterpolation": 0}, 0, 0, _P.PATH_MIXED_RADIX | _P.PATH_FUSED_BG),
    (1664, {"backgroundRemoval": 1, "rollingAverageWindowSize": 64, "resamplingInterpolation": 0}, 0, 0, _P.PATH_TEAM | _P.PATH_ROLL_IN_KERNEL),
    (1664, {"backgroundRemoval": 1, "rollingAverageWindowSize": 64}, 0, _P.ROUTE_NO_TEAM, _P.PATH_MIXED_RADIX | _P.PATH_PREPARED_ROWS),
    (1664, {"resamplingInterpolation": 2}, 0, 0, _P.PATH_MIXED_RADIX),
    (1000, {}, 0, 0, _P.PATH_MIXED_RADIX | _P.PATH_STATIC_PLAN),                               # the kernel compiled for the length at run time (10 x 10 x 10), raw uint16 rows
    (1000, {"backgroundRemoval": 1, "rollingAverageWindowSize": 64}, 0, 0, _P.PATH_MIXED_RADIX | _P.PATH_STATIC_PLAN | _P.PATH_ROLL_IN_KERNEL),
    (1000, {"backgroundRemoval": 1, "rollingAverageWindowSize": 300}, 0, 0, _P.PATH_MIXED_RADIX | _P.PATH_STATIC_PLAN | _P.PATH_PREPARED_ROWS),  # beyond the prefix-sum range
    (1000, {"backgroundRemoval": 1, "rollingAverageWindowSize": 64}, 0, _P.ROUTE_NO_MIXEDN_STATIC, _P.PATH_MIXED_RADIX | _P.PATH_PREPARED_ROWS),
    (1000, {"postProcessBackgroundRemoval": 1}, 0, 0, _P.PATH_MIXED_RADIX | _P.PATH_STATIC_PLAN | _P.PATH_FUSED_BG),
    (1000, {"bitDepth": 8}, 0, 0, _P.PATH_MIXED_RADIX | _P.PATH_STATIC_PLAN | _P.PATH_PREPARED_ROWS),
    (1000, {"dispersionCompensation": 0}, 0, 0, _P.PATH_MIXED_RADIX | _P.PATH_STATIC_PLAN | _P.PATH_REAL_INPUT),         # two A-scans per transform
    (1000, {"dispersionCompensation": 0}, 0, _P.ROUTE_NO_REAL_INPUT, _P.PATH_MIXED_RADIX | _P.PATH_STATIC_PLAN),
    (1000, {"dispersionCompensation": 0, "backgroundRemoval": 1, "rollingAverageWindowSize": 64}, 0, 0, _P.PATH_MIXED_RADIX | _P.PATH_STATIC_PLAN | _P.PATH_ROLL_IN_KERNEL),
    (1000, {"dispersionCompensation": 0, "bitDepth": 8}, 0, 0, _P.PATH_MIXED_RADIX | _P.PATH_STATIC_PLAN | _P.PATH_PREPARED_ROWS),
    (1000, {"resamplingInterpolation": 2}, 0, 0, _P.PATH_MIXED_RADIX | _P.PATH_STATIC_PLAN),      # Lanczos: the run-time compiled kernel too
    (1000, {"resamplingInterpolation": 2}, 0, _P.ROUTE_NO_MIXEDN_STATIC, _P.PATH_LIBRARY_FFT | _P.PATH_PREPARED_ROWS),  # ... not the run-time plan's
    (1000, {"resamplingInterpolation": 2, "backgroundRemoval": 1, "rollingAverageWindowSize": 16}, 0, 0, _P.PATH_MIXED_RADIX | _P.PATH_STATIC_PLAN | _P.PATH_PREPARED_ROWS),
    (1000, {}, 0, _P.ROUTE_NO_MIXEDN_STATIC, _P.PATH_MIXED_RADIX),                             # the run-time-plan kernel (mixedn_kernel.h)
    (1000, {"postProcessBackgroundRemoval": 1}, 0, _P.ROUTE_NO_MIXEDN_STATIC, _P.PATH_MIXED_RADIX | _P.PATH_FUSED_BG),
    (1000, {}, 0, _P.ROUTE_NO_MIXEDN, _P.PATH_LIBRARY_FFT | _P.PATH_PREPARED_ROWS),
    (1000, {}, 0, _P.ROUTE_NO_LIBFFT | _P.ROUTE_NO_MIXEDN, _P.PATH_BLUESTEIN | _P.PATH_PREPARED_ROWS),
    (2000, {}, 0, 0, _P.PATH_MIXED_RADIX | _P.PATH_STATIC_PLAN),
    (3000, {}, 0, 0, _P.PATH_MIXED_RADIX | _P.PATH_STATIC_PLAN),                               # 20 x 15 x 10: beyond the run-time plan's 2304
    (3000, {}, 0, _P.ROUTE_NO_MIXEDN_STATIC, _P.PATH_LIBRARY_FFT | _P.PATH_PREPARED_ROWS),
    (5000, {}, 0, 0, _P.PATH_MIXED_RADIX | _P.PATH_STATIC_PLAN),                               # 20 x 10 x 5 x 5: the longest lengths (<= 5120) run two A-scans per CU
    # 5120 < N <= 8192, even and 2-3-5-7-11-13-smooth (round 6): the same kernel on a TEAM of two waves per A-scan -- no hipFFT any more
    (6000, {}, 0, 0, _P.PATH_MIXED_RADIX | _P.PATH_STATIC_PLAN),                               # 15 x 20 x 20
    (6144, {"dispersionCompensation": 0}, 0, 0, _P.PATH_MIXED_RADIX | _P.PATH_STATIC_PLAN | _P.PATH_REAL_INPUT),
    (6144, {"resamplingInterpolation": 2}, 0, 0, _P.PATH_MIXED_RADIX | _P.PATH_STATIC_PLAN),   # Lanczos
    (6144, {"backgroundRemoval": 1, "rollingAverageWindowSize": 64}, 0, 0, _P.PATH_MIXED_RADIX | _P.PATH_STATIC_PLAN | _P.PATH_PREPARED_ROWS),  # the team has no rolling average inside
    (6144, {"postProcessBackgroundRemoval": 1}, 0, 0, _P.PATH_MIXED_RADIX | _P.PATH_STATIC_PLAN | _P.PATH_FUSED_BG),
    (8000, {}, 0, 0, _P.PATH_MIXED_RADIX | _P.PATH_STATIC_PLAN),
    (6000, {}, 0, _P.ROUTE_NO_MIXEDN_STATIC, _P.PATH_LIBRARY_FFT | _P.PATH_PREPARED_ROWS),     # ... the route of a process without hiprtc
    (6002, {}, 0, 0, _P.PATH_LIBRARY_FFT | _P.PATH_PREPARED_ROWS),                             # 2 x 3001: a prime factor above 13 -- the library route that is left
    (1234, {}, 0, 0, _P.PATH_LIBRARY_FFT | _P.PATH_PREPARED_ROWS),                             # 2 x 617: no plan
]


def row_id(v):
    return str(v).replace(" ", "") if not isinstance(v, dict) else ",".join("%s=%s" % kv for kv in v.items()) or "v180"
