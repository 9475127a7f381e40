// does ds_read_b128 work at 4-byte alignment on gfx950, and what does it cost against 2 x ds_read2_b32?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(const int* offs, float* out, int iters) {
	__shared__ float lds[8192 + 64];
	for (int i = threadIdx.x; i < 8192 + 64; i += 256) lds[i] = (float)i;
	__syncthreads();
	const int base = MODE == 2 ? (offs[threadIdx.x] & ~3) : offs[threadIdx.x];  // float index, arbitrary alignment (MODE 2: aligned)
	f32x4 acc = {0, 0, 0, 0};
	const uint32_t addr0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)(lds) + 4u * (uint32_t)base;
	for (int it = 0; it < iters; it++) {
		uint32_t addr = addr0 + 64u * (uint32_t)(it & 63);
		f32x4 v;
		if (MODE == 0 || MODE == 2) {
			asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
		} else {
			typedef float f32x2 __attribute__((ext_vector_type(2)));
			f32x2 lo, hi;
			asm volatile("ds_read2_b32 %0, %2 offset1:1\n\tds_read2_b32 %1, %2 offset0:2 offset1:3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(lo), "=&v"(hi) : "v"(addr) : "memory");
			v = f32x4{lo.x, lo.y, hi.x, hi.y};
		}
		acc += v;
	}
	out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y * 2 + acc.z * 3 + acc.w * 4;
}
int main() {
	std::vector<int> offs(256);
	for (int i = 0; i < 256; i++) offs[i] = (i * 17 + 3) % 4000;  // like resampled tap positions: irregular, unaligned
	int* d_offs; float* d_out;
	hipMalloc(&d_offs, 1024); hipMalloc(&d_out, 4 * 256 * 2048);
	hipMemcpy(d_offs, offs.data(), 1024, hipMemcpyHostToDevice);
	std::vector<float> r0(256), r1(256);
	hipLaunchKernelGGL(k<0>, dim3(1), dim3(256), 0, 0, d_offs, d_out, 1); hipMemcpy(r0.data(), d_out, 1024, hipMemcpyDeviceToHost);
	hipError_t e = hipDeviceSynchronize(); printf("b128 unaligned: %s\n", hipGetErrorString(e));
	hipLaunchKernelGGL(k<1>, dim3(1), dim3(256), 0, 0, d_offs, d_out, 1); hipMemcpy(r1.data(), d_out, 1024, hipMemcpyDeviceToHost);
	int bad = 0; for (int i = 0; i < 256; i++) if (r0[i] != r1[i]) bad++;
	printf("mismatches %d (r0[1]=%g r1[1]=%g)\n", bad, r0[1], r1[1]);
	for (int m = 0; m < 3; m++) {
		hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
		for (int rep = 0; rep < 2; rep++) {
			hipEventRecord(a);
			if (m == 0) hipLaunchKernelGGL(k<0>, dim3(2048), dim3(256), 0, 0, d_offs, d_out, 4096);
			else if (m == 1) hipLaunchKernelGGL(k<1>, dim3(2048), dim3(256), 0, 0, d_offs, d_out, 4096);
			else hipLaunchKernelGGL(k<2>, dim3(2048), dim3(256), 0, 0, d_offs, d_out, 4096);
			hipEventRecord(b); hipEventSynchronize(b);
			float ms; hipEventElapsedTime(&ms, a, b);
			if (rep) printf("mode %d: %.3f ms\n", m, ms);
		}
	}
	return 0;
}
