// micro-benchmark: issue cost of a few VALU ops on gfx950 (cycles per wave-instruction per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP 64
template <int OP> __global__ __launch_bounds__(256) void k(float* out, int iters, float s) {
	f2 a[8];
	for (int i = 0; i < 8; i++) a[i] = f2{s * (threadIdx.x + i), s * i};
	f2 c = f2{s, 1.0f + s};
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int r = 0; r < REP / 8; r++) {
#pragma unroll
			for (int i = 0; i < 8; i++) {
				if (OP == 0) { asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i].x) : "v"(c.x)); }
				if (OP == 1) { asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c)); }
				if (OP == 2) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(c.x)); }
				if (OP == 3) { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c)); }
				if (OP == 4) { asm volatile("v_mov_b32 %0, %1" : "+v"(a[i].x) : "v"(c.x)); }
				if (OP == 5) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c)); }
				if (OP == 6) { asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(c.x)); }
				if (OP == 7) { asm volatile("v_log_f32 %0, %0" : "+v"(a[i].x)); }
				if (OP == 8) { asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a[i].x)); }
				if (OP == 9) { asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "+v"(a[i]) : "v"(c)); }
				if (OP == 10) { asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i].x), "+v"(a[i].y)); }
				if (OP == 11) { asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[i].x), "+v"(a[i].y)); }
				if (OP == 12) { asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i].x) : "v"(a[i].y)); }
				if (OP == 13) { asm volatile("v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(a[i].x) : "v"(a[i].y)); }
			}
		}
	}
	float acc = 0;
	for (int i = 0; i < 8; i++) acc += a[i].x + a[i].y;
	out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int OP> void run(const char* name, int blocksPerCU) {
	float* d; hipMalloc(&d, 256 * 256 * 16 * 4);
	int iters = 2000;
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	int blocks = 256 * blocksPerCU;
	k<OP><<<blocks, 256>>>(d, 10, 0.5f); hipDeviceSynchronize();
	hipEventRecord(e0); k<OP><<<blocks, 256>>>(d, iters, 0.5f); hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	// per SIMD: blocksPerCU waves (each block = 4 waves over 4 SIMDs)
	double instPerSimd = (double)iters * REP * blocksPerCU;
	double cyc = ms * 1e-3 * 2.4e9 / instPerSimd;
	printf("%-28s waves/SIMD=%d  %.3f ms  %.2f cycles/inst (at 2.4 GHz)\n", name, blocksPerCU, ms, cyc);
	hipFree(d);
}
int main() {
	for (int w : {1, 2, 4}) {
		run<0>("v_fma_f32", w); run<1>("v_pk_fma_f32", w); run<2>("v_add_f32", w); run<3>("v_pk_add_f32", w);
		run<4>("v_mov_b32", w); run<5>("v_pk_mul_f32", w); run<6>("v_mul_f32", w); run<7>("v_log_f32", w);
		run<8>("v_cvt_f32_u32", w); run<9>("v_pk_add_f32 opsel/neg", w);
		run<10>("v_permlane32_swap", w); run<11>("v_permlane16_swap", w); run<12>("v_mov_dpp quad_perm", w); run<13>("v_mov_dpp row_ror:8", w);
	}
	return 0;
}
