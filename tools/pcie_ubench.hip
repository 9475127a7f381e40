// pcie_ubench.hip -- what the host link gives: H2D alone, D2H alone, both at once, for registered (hipHostRegister) and
// HIP-allocated (hipHostMalloc) host memory, 256 MiB transfers like one 1024 x 512 x 256 buffer.
//   hipcc --offload-arch=gfx950 -O2 tools/pcie_ubench.hip -o scratch/pcie_ubench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
	const size_t n = 256u << 20;
	void *dIn, *dOut;
	CK(hipMalloc(&dIn, n)); CK(hipMalloc(&dOut, n));
	hipStream_t s1, s2;
	CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
	for (int kind = 0; kind < 2; ++kind) {
		void *hIn, *hOut;
		if (kind == 0) {
			if (posix_memalign(&hIn, 4096, n) || posix_memalign(&hOut, 4096, n)) return 1;
			memset(hIn, 1, n); memset(hOut, 2, n);
			CK(hipHostRegister(hIn, n, hipHostRegisterPortable)); CK(hipHostRegister(hOut, n, hipHostRegisterPortable));
		} else {
			CK(hipHostMalloc(&hIn, n, hipHostMallocDefault)); CK(hipHostMalloc(&hOut, n, hipHostMallocDefault));
			memset(hIn, 1, n); memset(hOut, 2, n);
		}
		const char* name = kind == 0 ? "registered" : "hipHostMalloc";
		const int reps = 20;
		for (int mode = 0; mode < 3; ++mode) {
			for (int w = 0; w < 2; ++w) {  // pass 0 = warm-up
				const double t0 = now();
				for (int r = 0; r < reps; ++r) {
					if (mode != 1) CK(hipMemcpyAsync(dIn, hIn, n, hipMemcpyHostToDevice, s1));
					if (mode != 0) CK(hipMemcpyAsync(hOut, dOut, n, hipMemcpyDeviceToHost, s2));
				}
				CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
				const double dt = now() - t0;
				if (w) printf("%-14s %-10s %.2f ms per 256 MiB pair, %.1f GB/s per direction\n", name, mode == 0 ? "H2D" : mode == 1 ? "D2H" : "both", dt / reps * 1e3, n * reps / dt / 1e9);
			}
		}
	}
	return 0;
}
