// mfma_valu_ubench.hip -- do FP32 MFMAs of one wave run beside the FP32 VALU work of the OTHER wave of the same SIMD?
// 256 workgroups x 8 waves (two per SIMD: waves w and w + 4 share SIMD w % 4).  Role A (waves 0..3): a stream of independent
// v_mfma_f32_16x16x4_f32 (4 accumulator chains).  Role B (waves 4..7): a stream of independent packed / scalar FP32 VALU
// instructions.  Timed: A alone, B alone, A and B together.  If the two pipes are independent, together = max; if the f32
// matrix instruction occupies the vector ALU's FP32 datapath, together = sum.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_valu_ubench.hip -o scratch/mfma_valu_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); return 1; } } while (0)

template <int ROLES, int VKIND>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters) {
	const int wave = threadIdx.x >> 6;
	const bool roleA = wave < 4;
	float r = 0.0f;
	if (roleA) {
		if (ROLES & 1) {
			f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
			float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
			for (int i = 0; i < iters; i++) {
#pragma unroll
				for (int s = 0; s < 4; s++)
#pragma unroll
					for (int g = 0; g < 4; g++) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[g], 0, 0, 0);
			}
			r = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
		}
	} else if (ROLES & 2) {
		f2 x[8];
#pragma unroll
		for (int j = 0; j < 8; j++) x[j] = f2{threadIdx.x * 1e-3f + j, 1.0f + j};
		const f2 c = f2{1.0000001f, 0.9999999f}, d = f2{1e-7f, -1e-7f};
		for (int i = 0; i < iters; i++) {
#pragma unroll
			for (int rep = 0; rep < 8; rep++)
#pragma unroll
				for (int j = 0; j < 8; j++) {
					if (VKIND == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(c), "v"(d));
					else if (VKIND == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[j]) : "v"(d));
					else if (VKIND == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j].x) : "v"(c.x), "v"(d.x));
					else asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[j].x) : "v"(c.x));
				}
		}
		for (int j = 0; j < 8; j++) r += x[j].x + x[j].y;
	}
	if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int ROLES, int VKIND> double run(float* d, int iters) {
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL((k<ROLES, VKIND>), dim3(256), dim3(512), 0, 0, d, 100);
	hipDeviceSynchronize();
	hipEventRecord(e0);
	hipLaunchKernelGGL((k<ROLES, VKIND>), dim3(256), dim3(512), 0, 0, d, iters);
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms = 0; hipEventElapsedTime(&ms, e0, e1);
	return ms;
}
template <int VKIND> void report(float* d, const char* name) {
	const int iters = 20000;
	const double a = run<1, VKIND>(d, iters), b = run<2, VKIND>(d, iters), ab = run<3, VKIND>(d, iters);
	printf("%-14s MFMA alone %.3f ms (%.1f ns per MFMA), VALU alone %.3f ms (%.2f ns per instr), together %.3f ms  -> sum %.3f, max %.3f\n", name, a,
	       a * 1e6 / (iters * 16.0), b, b * 1e6 / (iters * 64.0), ab, a + b, a > b ? a : b);
}
int main() {
	float* d; CK(hipMalloc(&d, 4096));
	report<0>(d, "v_pk_fma_f32");
	report<1>(d, "v_pk_add_f32");
	report<2>(d, "v_fma_f32");
	report<3>(d, "v_xor_b32");
	return 0;
}
