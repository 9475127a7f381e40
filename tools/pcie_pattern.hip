// pcie_pattern.hip -- how long a 256 MiB device-to-host copy takes on a result stream that waits for a kernel of another
// stream (the pipeline's pattern), next to a concurrent H2D, under the HIP runtime the process happens to load
// (LD_LIBRARY_PATH=<torch>/lib runs it on the runtime PyTorch bundles).
//   hipcc --offload-arch=gfx950 -O2 tools/pcie_pattern.hip -o scratch/pcie_pattern
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(float* p, size_t n) {
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] += 1.0f;
}
int main(int argc, char** argv) {
	int ver = 0; CK(hipRuntimeGetVersion(&ver)); printf("HIP runtime %d\n", ver);
	const size_t n = 256u << 20;
	void *dIn, *dOut, *hIn, *hOut;
	CK(hipMalloc(&dIn, n)); CK(hipMalloc(&dOut, n));
	if (posix_memalign(&hIn, 4096, n) || posix_memalign(&hOut, 4096, n)) return 1;
	memset(hIn, 1, n); memset(hOut, 2, n);
	CK(hipHostRegister(hIn, n, hipHostRegisterPortable)); CK(hipHostRegister(hOut, n, hipHostRegisterPortable));
	hipStream_t sK, sOut, sIn;
	CK(hipStreamCreateWithFlags(&sK, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sOut, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sIn, hipStreamNonBlocking));
	hipEvent_t e, c0, c1;
	CK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); CK(hipEventCreate(&c0)); CK(hipEventCreate(&c1));
	const char* names[] = {"D2H alone", "kernel(sK) -> event -> wait(sOut) -> D2H", "same + concurrent H2D", "same, D2H split in 8 pieces + concurrent H2D",
	                       "kernel on sOut itself -> D2H + concurrent H2D", "event wait -> tiny D2H first -> D2H + concurrent H2D"};
	for (int mode = 0; mode < 6; ++mode) {
		double wall = 0, copy = 0;
		const int reps = 12;
		for (int r = 0; r < reps + 2; ++r) {
			const double t0 = now();
			if (mode == 4) {
				hipLaunchKernelGGL(touch, dim3(1024), dim3(256), 0, sOut, (float*)dOut, n / 4);
			} else if (mode >= 1) {
				hipLaunchKernelGGL(touch, dim3(1024), dim3(256), 0, sK, (float*)dOut, n / 4);
				CK(hipEventRecord(e, sK));
				CK(hipStreamWaitEvent(sOut, e, 0));
			}
			if (mode >= 2) CK(hipMemcpyAsync(dIn, hIn, n, hipMemcpyHostToDevice, sIn));
			if (mode == 5) CK(hipMemcpyAsync(hOut, dOut, 64, hipMemcpyDeviceToHost, sOut));
			CK(hipEventRecord(c0, sOut));
			if (mode == 3) { for (int k = 0; k < 8; ++k) CK(hipMemcpyAsync((char*)hOut + k * (n / 8), (char*)dOut + k * (n / 8), n / 8, hipMemcpyDeviceToHost, sOut)); }
			else CK(hipMemcpyAsync(hOut, dOut, n, hipMemcpyDeviceToHost, sOut));
			CK(hipEventRecord(c1, sOut));
			CK(hipStreamSynchronize(sOut)); CK(hipStreamSynchronize(sIn)); CK(hipStreamSynchronize(sK));
			float ms = 0; CK(hipEventElapsedTime(&ms, c0, c1));
			if (r >= 2) { wall += now() - t0; copy += ms; }
		}
		printf("%-58s D2H %.2f ms, iteration %.2f ms\n", names[mode], copy / reps, wall / reps * 1e3);
	}
	// the pipeline's steady state: no host synchronisation except the wait for the H2D copy; with / without a host function
	// behind every D2H copy
	hipEvent_t h2d[2], free_[2];
	for (int i = 0; i < 2; ++i) { CK(hipEventCreateWithFlags(&h2d[i], hipEventBlockingSync | hipEventDisableTiming)); CK(hipEventCreateWithFlags(&free_[i], hipEventDisableTiming)); }
	for (int withFn = 0; withFn < 3; ++withFn) {
		const int reps = 40;
		double t0 = 0;
		for (int r = 0; r < reps + 4; ++r) {
			if (r == 4) t0 = now();
			CK(hipMemcpyAsync(dIn, hIn, n, hipMemcpyHostToDevice, sIn));
			CK(hipEventRecord(h2d[r & 1], sIn));
			CK(hipStreamWaitEvent(sK, h2d[r & 1], 0));
			hipLaunchKernelGGL(touch, dim3(1024), dim3(256), 0, sK, (float*)dOut, n / 4);
			CK(hipEventRecord(e, sK));
			if (withFn == 2) {  // the D2H copy is enqueued only once its input is complete (what a result thread would do)
				CK(hipEventSynchronize(h2d[r & 1]));
				CK(hipEventSynchronize(e));
				CK(hipMemcpyAsync(hOut, dOut, n, hipMemcpyDeviceToHost, sOut));
				continue;
			}
			CK(hipStreamWaitEvent(sOut, e, 0));
			CK(hipMemcpyAsync(hOut, dOut, n, hipMemcpyDeviceToHost, sOut));
			if (withFn) CK(hipLaunchHostFunc(sOut, [](void*) {}, nullptr));
			CK(hipEventSynchronize(h2d[r & 1]));
		}
		CK(hipStreamSynchronize(sOut)); CK(hipStreamSynchronize(sIn)); CK(hipStreamSynchronize(sK));
		printf("pipelined loop, host function behind every D2H: %s  %.2f ms per buffer\n", withFn == 2 ? "no, D2H enqueued after its input completed" : withFn ? "yes" : "no", (now() - t0) / reps * 1e3);
	}
	return 0;
}
