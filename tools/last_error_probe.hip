// last_error_probe.hip -- which successful HIP calls leave a status in the thread's last-error slot?  (Found in round 4: a later
// hipGetLastError() -- PyTorch's launch check, or this library's own launchers -- reported "pointer does not correspond to a
// registered memory region" although every call had returned hipSuccess.)
//   hipcc --offload-arch=gfx950 -O2 tools/last_error_probe.hip -o scratch/last_error_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define SHOW(what, call) do { hipError_t r_ = (call); hipError_t p_ = hipPeekAtLastError(); \
	printf("%-70s returned %-12s last-error slot: %s\n", what, hipGetErrorName(r_), hipGetErrorName(p_)); (void)hipGetLastError(); } while (0)
__global__ void k(float* p) { p[threadIdx.x] = 1.0f; }
int main() {
	int ver = 0; hipRuntimeGetVersion(&ver); printf("HIP runtime %d\n", ver);
	hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
	void* d; hipMalloc(&d, 64 << 20);
	(void)hipGetLastError();
	for (size_t bytes : {size_t(256), size_t(16) << 10, size_t(64) << 10, size_t(1) << 20, size_t(32) << 20}) {
		std::vector<char> host(bytes + 64, 1);
		char what[128];
		snprintf(what, sizeof what, "hipMemcpyAsync H2D from pageable memory, %zu bytes", bytes);
		SHOW(what, hipMemcpyAsync(d, host.data() + 8, bytes, hipMemcpyHostToDevice, s));
		SHOW("  hipStreamSynchronize", hipStreamSynchronize(s));
		snprintf(what, sizeof what, "hipMemcpyAsync D2H into pageable memory, %zu bytes", bytes);
		SHOW(what, hipMemcpyAsync(host.data() + 8, d, bytes, hipMemcpyDeviceToHost, s));
		SHOW("  hipStreamSynchronize", hipStreamSynchronize(s));
		snprintf(what, sizeof what, "hipMemcpy H2D (blocking) from pageable memory, %zu bytes", bytes);
		SHOW(what, hipMemcpy(d, host.data() + 8, bytes, hipMemcpyHostToDevice));
	}
	std::vector<char> host(1 << 20, 1);
	hipPointerAttribute_t attr{};
	SHOW("hipPointerGetAttributes on pageable memory", hipPointerGetAttributes(&attr, host.data()));
	SHOW("hipHostRegister", hipHostRegister(host.data(), host.size(), hipHostRegisterPortable));
	SHOW("hipMemcpyAsync H2D from registered memory", hipMemcpyAsync(d, host.data(), host.size(), hipMemcpyHostToDevice, s));
	SHOW("hipHostUnregister", hipHostUnregister(host.data()));
	SHOW("hipStreamQuery (idle stream)", hipStreamQuery(s));
	hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s, (float*)d);
	SHOW("hipStreamQuery (busy or just finished)", hipStreamQuery(s));
	SHOW("hipMemsetAsync", hipMemsetAsync(d, 0, 1 << 20, s));
	SHOW("hipStreamSynchronize", hipStreamSynchronize(s));
	SHOW("hipFree", hipFree(d));
	return 0;
}
