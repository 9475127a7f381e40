// rtc_probe.hip -- what a kernel compiled at run time (hiprtc) may use of the 160 KiB of LDS when it is launched through
// hipModuleLaunchKernel: (a) a static __shared__ array of 152 KiB, (b) 152 KiB of dynamic LDS without any attribute,
// (c) dynamic LDS after hipFuncSetAttribute on the hipFunction_t.  Prints which of the three launch and return the right values.
//   hipcc --offload-arch=gfx950 -O2 tools/rtc_probe.hip -o scratch/rtc_probe -lhiprtc
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <cstdio>
#include <string>
#include <vector>
#include <chrono>

static const char* kSrc = R"(
extern "C" __global__ void k_static(float* o) {
	__shared__ float s[38912];
	for (int i = threadIdx.x; i < 38912; i += blockDim.x) s[i] = (float)i;
	__syncthreads();
	o[threadIdx.x] = s[38911 - threadIdx.x];
}
extern "C" __global__ void k_dynamic(float* o) {
	extern __shared__ float d[];
	for (int i = threadIdx.x; i < 38912; i += blockDim.x) d[i] = (float)i;
	__syncthreads();
	o[threadIdx.x] = d[38911 - threadIdx.x];
}
)";

static bool check(const char* what, hipError_t e, float* d_o) {
	if (e != hipSuccess) { printf("%-58s launch error: %s\n", what, hipGetErrorString(e)); (void)hipGetLastError(); return false; }
	e = hipDeviceSynchronize();
	if (e != hipSuccess) { printf("%-58s sync error: %s\n", what, hipGetErrorString(e)); (void)hipGetLastError(); return false; }
	std::vector<float> h(256);
	(void)hipMemcpy(h.data(), d_o, 1024, hipMemcpyDeviceToHost);
	bool ok = true;
	for (int i = 0; i < 256; ++i) ok = ok && h[i] == (float)(38911 - i);
	printf("%-58s %s\n", what, ok ? "ok" : "WRONG VALUES");
	(void)hipMemset(d_o, 0, 1024);
	return ok;
}

int main() {
	int ver = 0; (void)hipRuntimeGetVersion(&ver);
	hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
	printf("HIP runtime %d, %s, sharedMemPerBlock %zu, maxSharedMemoryPerMultiProcessor %zu\n", ver, prop.gcnArchName, prop.sharedMemPerBlock, prop.maxSharedMemoryPerMultiProcessor);
	hiprtcProgram prog;
	hiprtcCreateProgram(&prog, kSrc, "probe.hip", 0, nullptr, nullptr);
	std::string arch = std::string("--offload-arch=") + prop.gcnArchName;
	const char* opts[] = {arch.c_str(), "-O3"};
	auto t0 = std::chrono::steady_clock::now();
	hiprtcResult r = hiprtcCompileProgram(prog, 2, opts);
	printf("hiprtcCompileProgram: %s (%.2f s)\n", hiprtcGetErrorString(r), std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
	if (r != HIPRTC_SUCCESS) { size_t n = 0; hiprtcGetProgramLogSize(prog, &n); std::string log(n, 0); hiprtcGetProgramLog(prog, &log[0]); printf("%s\n", log.c_str()); return 1; }
	size_t cs = 0; hiprtcGetCodeSize(prog, &cs);
	std::vector<char> code(cs); hiprtcGetCode(prog, code.data());
	hipModule_t mod; hipFunction_t fs, fd;
	if (hipModuleLoadData(&mod, code.data()) != hipSuccess) { printf("hipModuleLoadData failed\n"); return 1; }
	(void)hipModuleGetFunction(&fs, mod, "k_static");
	(void)hipModuleGetFunction(&fd, mod, "k_dynamic");
	float* d_o; (void)hipMalloc(&d_o, 1024); (void)hipMemset(d_o, 0, 1024);
	void* args[] = {&d_o};
	check("(a) static 152 KiB, hipModuleLaunchKernel", hipModuleLaunchKernel(fs, 1, 1, 1, 256, 1, 1, 0, nullptr, args, nullptr), d_o);
	check("(b) dynamic 152 KiB, no attribute", hipModuleLaunchKernel(fd, 1, 1, 1, 256, 1, 1, 38912 * 4, nullptr, args, nullptr), d_o);
	hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(fd), hipFuncAttributeMaxDynamicSharedMemorySize, 38912 * 4);
	printf("hipFuncSetAttribute(hipFunction_t): %s\n", hipGetErrorString(ea)); (void)hipGetLastError();
	check("(c) dynamic 152 KiB after hipFuncSetAttribute", hipModuleLaunchKernel(fd, 1, 1, 1, 256, 1, 1, 38912 * 4, nullptr, args, nullptr), d_o);
	int v = 0;
	printf("hipFuncGetAttribute(SHARED_SIZE_BYTES, k_static): %s -> %d\n", hipGetErrorString(hipFuncGetAttribute(&v, HIP_FUNC_ATTRIBUTE_SHARED_SIZE_BYTES, fs)), v);
	return 0;
}
