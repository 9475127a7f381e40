// pcie_kernel_copy.hip -- result delivery by a COPY KERNEL that writes straight into registered host memory (zero-copy over PCIe)
// instead of hipMemcpyAsync device-to-host on the runtime's DMA engines, next to the concurrent DMA host-to-device copy of the
// next raw buffer: does it reach the link's both-directions rate under the HIP runtime the process happens to run on?
// (tools/pcie_pattern.hip showed that under the runtime PyTorch bundles, 7.0.2, a pipelined H2D + D2H loop of DMA copies overlaps
// only partially: 7.2-9.7 ms per 256 MiB pair against 5.7 ms on the system's 7.2 runtime.)
//   hipcc --offload-arch=gfx950 -O2 tools/pcie_kernel_copy.hip -o scratch/pcie_kernel_copy
//   scratch/pcie_kernel_copy [workgroups of the copy kernel = 64]
//   LD_PRELOAD=<torch>/lib/libamdhip64.so scratch/pcie_kernel_copy      (the bundled runtime)
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
// 16 bytes per lane and iteration, consecutive lanes consecutive addresses: 1 KiB per wave instruction towards the host
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void copy_out(f32x4* __restrict__ dst, const f32x4* __restrict__ src, size_t n16) {
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) __builtin_nontemporal_store(src[i], &dst[i]);
}

int main(int argc, char** argv) {
	int ver = 0; CK(hipRuntimeGetVersion(&ver)); printf("HIP runtime %d\n", ver);
	const int wgs = argc > 1 ? atoi(argv[1]) : 64;
	const size_t n = 256u << 20;
	void *dIn[2], *dOut[2], *hIn, *hOut[2], *hOutDev[2];
	for (int i = 0; i < 2; ++i) { CK(hipMalloc(&dIn[i], n)); CK(hipMalloc(&dOut[i], n)); CK(hipMemset(dOut[i], 0, n)); }
	if (posix_memalign(&hIn, 4096, n)) return 1;
	memset(hIn, 1, n);
	CK(hipHostRegister(hIn, n, hipHostRegisterPortable));
	for (int i = 0; i < 2; ++i) {
		if (posix_memalign(&hOut[i], 4096, n)) return 1;
		memset(hOut[i], 2, n);
		CK(hipHostRegister(hOut[i], n, hipHostRegisterPortable | hipHostRegisterMapped));
		CK(hipHostGetDevicePointer(&hOutDev[i], hOut[i], 0));
	}
	int least = 0, greatest = 0;
	CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
	hipStream_t sK, sOut, sIn;
	CK(hipStreamCreateWithFlags(&sK, hipStreamNonBlocking));
	CK(hipStreamCreateWithPriority(&sIn, hipStreamNonBlocking, greatest));
	CK(hipStreamCreateWithPriority(&sOut, hipStreamNonBlocking, least));
	hipEvent_t h2d[2], done[2], read[2];
	for (int i = 0; i < 2; ++i) { CK(hipEventCreateWithFlags(&h2d[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&done[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&read[i], hipEventDisableTiming)); }
	CK(hipDeviceSynchronize());

	for (int mode = 0; mode < 4; ++mode) {
		// 0: kernel copy alone   1: DMA D2H alone   2: pipeline, results by the copy kernel   3: pipeline, results by DMA (hipMemcpyAsync)
		const int reps = 40;
		double t0 = 0;
		for (int r = 0; r < reps + 4; ++r) {
			if (r == 4) { CK(hipDeviceSynchronize()); t0 = now(); }
			const int s = r & 1;
			if (mode == 0) {
				hipLaunchKernelGGL(copy_out, dim3(wgs), dim3(256), 0, sOut, (f32x4*)hOutDev[s], (const f32x4*)dOut[s], n / 16);
			} else if (mode == 1) {
				CK(hipMemcpyAsync(hOut[s], dOut[s], n, hipMemcpyDeviceToHost, sOut));
			} else {
				// the pipeline's shape: H2D of buffer r (copy stream) -> kernel (compute stream) -> result out (result stream); the result of
				// buffer r goes out while the H2D of buffer r + 1 comes in
				if (r >= 2) CK(hipStreamWaitEvent(sIn, done[s], 0));    // raw slot free again
				CK(hipMemcpyAsync(dIn[s], hIn, n, hipMemcpyHostToDevice, sIn));
				CK(hipEventRecord(h2d[s], sIn));
				CK(hipStreamWaitEvent(sK, h2d[s], 0));
				if (r >= 2) CK(hipStreamWaitEvent(sK, read[s], 0));      // result slot read out
				hipLaunchKernelGGL(touch, dim3(1024), dim3(256), 0, sK, (float*)dOut[s], n / 4);
				CK(hipEventRecord(done[s], sK));
				CK(hipStreamWaitEvent(sOut, done[s], 0));
				if (mode == 2) hipLaunchKernelGGL(copy_out, dim3(wgs), dim3(256), 0, sOut, (f32x4*)hOutDev[s], (const f32x4*)dOut[s], n / 16);
				else CK(hipMemcpyAsync(hOut[s], dOut[s], n, hipMemcpyDeviceToHost, sOut));
				CK(hipEventRecord(read[s], sOut));
			}
		}
		CK(hipDeviceSynchronize());
		const double ms = (now() - t0) / reps * 1e3;
		const char* names[] = {"copy kernel to host alone", "DMA D2H alone", "pipeline H2D(DMA) + kernel + results by copy kernel", "pipeline H2D(DMA) + kernel + results by DMA D2H"};
		printf("%-56s %.2f ms per 256 MiB buffer (%.1f GB/s per direction)", names[mode], ms, n / ms / 1e6);
		if (mode == 0 || mode == 2) printf("  [%d workgroups]", wgs);
		if (mode == 2) {  // the copied data is what the device holds (slot of the last iteration)
			const int s = (reps + 3) & 1;
			void* chk = malloc(n);
			CK(hipMemcpy(chk, dOut[s], n, hipMemcpyDeviceToHost));
			printf("  last buffer %s", memcmp(chk, hOut[s], n) == 0 ? "verified" : "DIFFERS");
			free(chk);
		}
		printf("\n");
	}
	return 0;
}
