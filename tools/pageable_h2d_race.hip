// pageable_h2d_race.hip -- does the HIP runtime deliver concurrent hipMemcpyAsync H2D copies out of ONE pageable host buffer
// correctly when several host threads copy adjacent slabs of it, each on a non-blocking stream of its own, and a kernel on a
// second stream of the same thread consumes the slab behind an event?  That is the access pattern of octpipe_group_process with
// one submitting thread per member and a caller-owned, unpinned buffer (csrc/octpipe_group.hip); this program reproduces it
// without the library (root cause of the one wrong image seen on that path).
//
//   hipcc --offload-arch=gfx950 -O2 -pthread tools/pageable_h2d_race.hip -o /tmp/h2d_race
//   /tmp/h2d_race [threads=4] [slabKiB=512] [iterations=2000] [mode: 0 pageable, 1 registered]
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ void sum_kernel(const uint16_t* in, size_t n, unsigned long long* out) {
	unsigned long long s = 0;
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += (unsigned long long)in[i] * (unsigned long long)((i & 1023) + 1);
	atomicAdd(out, s);
}

int main(int argc, char** argv) {
	const int T = argc > 1 ? atoi(argv[1]) : 4;
	const size_t slab = (size_t)(argc > 2 ? atoi(argv[2]) : 512) * 1024 / 2;  // uint16 samples per slab
	const int iters = argc > 3 ? atoi(argv[3]) : 2000;
	const int mode = argc > 4 ? atoi(argv[4]) : 0;
	CK(hipSetDevice(0));
	std::vector<hipStream_t> copyS(T), compS(T);
	std::vector<hipEvent_t> ev(T);
	std::vector<uint16_t*> d(T);
	std::vector<unsigned long long*> dsum(T);
	for (int t = 0; t < T; ++t) {
		CK(hipStreamCreateWithFlags(&copyS[t], hipStreamNonBlocking));
		CK(hipStreamCreateWithFlags(&compS[t], hipStreamNonBlocking));
		CK(hipEventCreateWithFlags(&ev[t], hipEventDisableTiming));
		CK(hipMalloc(&dsum[t], 8));
	}
	std::atomic<int> bad{0};
	uint32_t seed = 12345u;
	for (int it = 0; it < iters; ++it) {
		const size_t n = slab * T;
		uint16_t* base = (uint16_t*)malloc((n + 4096) * 2);
		seed = seed * 1664525u + 1013904223u;
		const size_t off = ((seed >> 8) % 2040) | 1;  // odd sample offset: slabs share pages at their boundaries
		uint16_t* h = base + off;
		for (size_t i = 0; i < n; ++i) h[i] = (uint16_t)((i * 2654435761u + (uint32_t)it * 40503u) >> 13);
		if (mode == 1) CK(hipHostRegister(h, n * 2, hipHostRegisterPortable));
		std::vector<unsigned long long> want(T, 0), got(T, 0);
		for (int t = 0; t < T; ++t)
			for (size_t i = 0; i < slab; ++i) want[t] += (unsigned long long)h[t * slab + i] * (unsigned long long)((i & 1023) + 1);
		std::vector<std::thread> th;
		for (int t = 0; t < T; ++t)
			th.emplace_back([&, t] {
				CK(hipSetDevice(0));
				// fresh device slab per iteration, zero-filled on the null stream like the library's ensure()
				CK(hipMalloc(&d[t], slab * 2));
				CK(hipMemset(d[t], 0, slab * 2));
				CK(hipStreamSynchronize(nullptr));
				CK(hipMemsetAsync(dsum[t], 0, 8, compS[t]));
				CK(hipMemcpyAsync(d[t], h + t * slab, slab * 2, hipMemcpyHostToDevice, copyS[t]));
				CK(hipEventRecord(ev[t], copyS[t]));
				CK(hipStreamWaitEvent(compS[t], ev[t], 0));
				hipLaunchKernelGGL(sum_kernel, dim3(64), dim3(256), 0, compS[t], d[t], slab, dsum[t]);
				CK(hipMemcpyAsync(&got[t], dsum[t], 8, hipMemcpyDeviceToHost, compS[t]));
				CK(hipStreamSynchronize(compS[t]));
			});
		for (auto& x : th) x.join();
		for (int t = 0; t < T; ++t) {
			if (got[t] != want[t]) {
				// what does the device slab hold?
				std::vector<uint16_t> back(slab);
				CK(hipMemcpy(back.data(), d[t], slab * 2, hipMemcpyDeviceToHost));
				size_t diff = 0, first = slab, zeros = 0;
				for (size_t i = 0; i < slab; ++i) if (back[i] != h[t * slab + i]) { if (first == slab) first = i; ++diff; if (back[i] == 0) ++zeros; }
				printf("iteration %d thread %d: kernel saw a different slab (checksum); device slab now differs in %zu samples (first %zu, %zu of them zero)\n", it, t, diff, first, zeros);
				bad++;
			}
			CK(hipFree(d[t]));
		}
		if (mode == 1) CK(hipHostUnregister(h));
		free(base);
	}
	printf("threads %d, slab %zu KiB, %d iterations, mode %s: %d wrong slabs\n", T, slab * 2 / 1024, iters, mode ? "registered" : "pageable", bad.load());
	return bad.load() ? 1 : 0;
}
