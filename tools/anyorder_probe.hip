// anyorder_probe.hip -- does hipExtAnyOrderLaunch let a kernel start while the previous kernel of the SAME stream is still running on gfx950?
// (hip_ext.h notes the flag as "not supported on AMD GFX9xx boards"; this measures what the runtime of this image does.)
//   hipcc --offload-arch=gfx950 -O2 tools/anyorder_probe.hip -o scratch/anyorder_probe && scratch/anyorder_probe
// Kernel A spins ~1 ms on ONE workgroup, kernel B (any-order or ordinary) spins ~1 ms on one workgroup too: overlapped -> ~1 ms, ordered -> ~2 ms.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <chrono>

__global__ void spin(long long cycles, unsigned long long* out) {
	const long long t0 = wall_clock64();
	while (wall_clock64() - t0 < cycles) {}
	if (threadIdx.x == 0) out[blockIdx.x] = (unsigned long long)wall_clock64();
}

int main() {
	hipStream_t s;
	if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return 1;
	unsigned long long* d = nullptr;
	if (hipMalloc((void**)&d, 64) != hipSuccess) return 1;
	int rate = 0;
	hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);  // kHz
	const long long cyc = (long long)rate;  // 1 ms
	for (int mode = 0; mode < 2; mode++) {
		for (int rep = 0; rep < 3; rep++) {
			hipStreamSynchronize(s);
			const auto t0 = std::chrono::steady_clock::now();
			hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, 0u, cyc, d);
			hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, mode ? (unsigned)hipExtAnyOrderLaunch : 0u, cyc, d + 1);
			hipStreamSynchronize(s);
			const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
			unsigned long long h[2];
			hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
			printf("%s: two 1 ms kernels on one stream took %.3f ms; B finished %.3f ms after A\n", mode ? "any-order" : "ordered  ", ms, (double)((long long)h[1] - (long long)h[0]) / rate);
		}
	}
	printf("status %s\n", hipGetErrorString(hipGetLastError()));
	return 0;
}
