// Calibration of rocprofv3's WRITE_SIZE on gfx950 for the store patterns of this library (MI355X_MICROARCH.md: exact for
// 16-B-per-lane streaming stores, "other access widths are uncalibrated").  Each kernel stores exactly N·4 bytes:
//   k_dword_rows   one dword per lane, 64 consecutive floats per wave (a row segment of F: 256 contiguous bytes)
//   k_float4       16 bytes per lane, streaming
//   k_dword_runs16 one dword per lane, the wave's 64 floats as four runs of 16 at scattered 64-byte-aligned places
//   k_rec24        24-byte hit records, one per lane, consecutive (k_trace_rays' output)
// build: hipcc -O3 --offload-arch=gfx950 -o tools/ub/write_calib tools/ub/write_calib.hip
// run:   rocprofv3 --pmc WRITE_SIZE -- tools/ub/write_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k_dword_rows(float *p, uint64_t n) {
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) p[i] = (float)i;
}
__global__ void k_float4(float4 *p, uint64_t n4) {
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n4) p[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
__global__ void k_dword_runs16(float *p, uint64_t n) { // wave w, lane l: run (l >> 4) of the wave goes to a permuted 16-float slot
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint64_t slot = i >> 4, slots = n >> 4;
	const uint64_t perm = (slot * 2654435761ull + 12345ull) % slots; // a bijection when slots is a power of two (odd multiplier)
	p[perm * 16 + (i & 15)] = (float)i;
}
struct Rec { double t; float u, v; uint32_t a, b; };
__global__ void k_rec24(Rec *p, uint64_t n) {
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) { Rec r = {(double)i, 1.f, 2.f, (uint32_t)i, 7u}; p[i] = r; }
}
int main() {
	const uint64_t n = 1ull << 26; // 256 MiB of floats
	float *p = nullptr;
	if (hipMalloc(&p, n * 4 * 6) != hipSuccess) return 1;
	hipLaunchKernelGGL(k_dword_rows, dim3(n / 256), dim3(256), 0, 0, p, n);
	hipLaunchKernelGGL(k_float4, dim3(n / 4 / 256), dim3(256), 0, 0, (float4 *)p, n / 4);
	hipLaunchKernelGGL(k_dword_runs16, dim3(n / 256), dim3(256), 0, 0, p, n);
	hipLaunchKernelGGL(k_rec24, dim3(n / 256), dim3(256), 0, 0, (Rec *)p, n);
	if (hipDeviceSynchronize() != hipSuccess) return 2;
	printf("stored bytes: k_dword_rows %llu, k_float4 %llu, k_dword_runs16 %llu, k_rec24 %llu\n", (unsigned long long)n * 4, (unsigned long long)n * 4,
	       (unsigned long long)n * 4, (unsigned long long)n * 24);
	(void)hipFree(p);
	return 0;
}
