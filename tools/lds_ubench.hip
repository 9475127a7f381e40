// micro-benchmark: LDS-pipe cost of the DS instruction forms the fused kernel uses (cycles per
// wave-instruction per CU), 16 waves per CU, 16 DS operations per s_waitcnt.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define R16(X) X X X X X X X X X X X X X X X X
template <int OP> __global__ __launch_bounds__(1024) void k(float* out, int iters) {
	extern __shared__ char smem[];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	// per-wave slice of 8 KiB; lane-contiguous addressing (conflict-free)
	unsigned a32 = wave * 8192 + lane * 4, a64 = wave * 8192 + lane * 8, a128 = wave * 8192 + lane * 16;
	f2 v2 = f2{(float)lane, 1.0f};
	f4 v4 = f4{(float)lane, 1.0f, 2.0f, 3.0f};
	float v1 = (float)lane;
	f4 r4 = v4; f2 r2 = v2; float r1 = v1;
	for (int it = 0; it < iters; it++) {
		if (OP == 0) { R16(asm volatile("ds_read_b32 %0, %1 offset:256" : "=v"(r1) : "v"(a32));) }
		if (OP == 1) { R16(asm volatile("ds_read_b64 %0, %1 offset:512" : "=v"(r2) : "v"(a64));) }
		if (OP == 2) { R16(asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(r4) : "v"(a128));) }
		if (OP == 3) { R16(asm volatile("ds_read2_b32 %0, %1 offset0:64 offset1:128" : "=v"(r2) : "v"(a32));) }
		if (OP == 4) { R16(asm volatile("ds_read2_b64 %0, %1 offset0:64 offset1:128" : "=v"(r4) : "v"(a64));) }
		if (OP == 5) { R16(asm volatile("ds_read2st64_b64 %0, %1 offset0:1 offset1:2" : "=v"(r4) : "v"(a64));) }
		if (OP == 6) { R16(asm volatile("ds_write_b32 %0, %1 offset:256" : : "v"(a32), "v"(v1));) }
		if (OP == 7) { R16(asm volatile("ds_write_b64 %0, %1 offset:512" : : "v"(a64), "v"(v2));) }
		if (OP == 8) { R16(asm volatile("ds_write_b128 %0, %1 offset:1024" : : "v"(a128), "v"(v4));) }
		if (OP == 9) { R16(asm volatile("ds_write2_b64 %0, %1, %2 offset0:64 offset1:128" : : "v"(a64), "v"(v2), "v"(r2));) }
		if (OP == 10) { R16(asm volatile("ds_write2_b32 %0, %1, %2 offset0:64 offset1:128" : : "v"(a32), "v"(v1), "v"(r1));) }
		if (OP == 11) { R16(asm volatile("ds_read_b64 %0, %1 offset:516" : "=v"(r2) : "v"(a64));) }      // 4-byte aligned only
		if (OP == 12) { R16(asm volatile("ds_read_b128 %0, %1 offset:1028" : "=v"(r4) : "v"(a128));) }   // 4-byte aligned only
		if (OP == 13) { R16(asm volatile("ds_read_b128 %0, %1 offset:1028" : "=v"(r4) : "v"(a32));) }    // taps: lane stride 4 B, unaligned
		if (OP == 14) { R16(asm volatile("ds_read_b64 %0, %1 offset:516" : "=v"(r2) : "v"(a32));) }      // tap pairs: lane stride 4 B
		if (OP == 15) { R16(asm volatile("ds_read2_b32 %0, %1 offset0:1 offset1:2" : "=v"(r2) : "v"(a32));) }  // what hipcc emits for taps
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = r1 + r2.x + r4.x;
}
template <int OP> void run(const char* name, int bytesPerLane) {
	float* d; hipMalloc(&d, 256 * 1024 * 4);
	int iters = 4000;
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipFuncSetAttribute(reinterpret_cast<const void*>(k<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, 16 * 8192 + 4096);
	k<OP><<<256, 1024, 16 * 8192 + 4096>>>(d, 10); hipDeviceSynchronize();
	hipEventRecord(e0); k<OP><<<256, 1024, 16 * 8192 + 4096>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	double instPerCU = (double)iters * 16 * 16;  // 16 waves x 16 instructions
	double cyc = ms * 1e-3 * 2.4e9 / instPerCU;
	printf("%-20s %.3f ms  %.2f cycles/wave-instr/CU (at 2.4 GHz)  %.0f B/clk/CU\n", name, ms, cyc, 64.0 * bytesPerLane / cyc);
	hipFree(d);
}
// correctness of the unaligned forms: LDS word i holds (float)i
__global__ void kcheck(float* out) {
	__shared__ float s[1024];
	for (int i = threadIdx.x; i < 1024; i += 64) s[i] = (float)i;
	__syncthreads();
	unsigned a = threadIdx.x * 4;
	f4 r4; f2 r2;
	asm volatile("ds_read_b128 %0, %1 offset:4\n s_waitcnt lgkmcnt(0)" : "=v"(r4) : "v"(a) : "memory");
	asm volatile("ds_read_b64 %0, %1 offset:4\n s_waitcnt lgkmcnt(0)" : "=v"(r2) : "v"(a) : "memory");
	out[threadIdx.x * 6 + 0] = r4.x; out[threadIdx.x * 6 + 1] = r4.y; out[threadIdx.x * 6 + 2] = r4.z; out[threadIdx.x * 6 + 3] = r4.w;
	out[threadIdx.x * 6 + 4] = r2.x; out[threadIdx.x * 6 + 5] = r2.y;
}
void check() {
	float* d; hipMalloc(&d, 64 * 6 * 4); float h[64 * 6];
	kcheck<<<1, 64>>>(d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
	int bad = 0;
	for (int l = 0; l < 64; l++) {
		for (int j = 0; j < 4; j++) bad += h[l * 6 + j] != (float)(l + 1 + j);
		for (int j = 0; j < 2; j++) bad += h[l * 6 + 4 + j] != (float)(l + 1 + j);
	}
	printf("unaligned b128/b64 reads (lane stride 4 B): %s  lane5: %g %g %g %g | %g %g\n", bad ? "WRONG" : "correct", h[30], h[31], h[32], h[33], h[34], h[35]);
}
int main() {
	run<0>("ds_read_b32", 4); run<1>("ds_read_b64", 8); run<2>("ds_read_b128", 16); run<3>("ds_read2_b32", 8);
	run<4>("ds_read2_b64", 16); run<5>("ds_read2st64_b64", 16); run<6>("ds_write_b32", 4); run<7>("ds_write_b64", 8);
	run<8>("ds_write_b128", 16); run<9>("ds_write2_b64", 16); run<10>("ds_write2_b32", 8);
	run<11>("ds_read_b64 unaligned", 8); run<12>("ds_read_b128 unaligned", 16); run<13>("ds_read_b128 taps", 16);
	run<14>("ds_read_b64 taps", 8); run<15>("ds_read2_b32 taps", 8);
	check();
	return 0;
}
