// Issue rate of the vector-ALU instructions the traversal kernels are made of (gfx950): cycles a SIMD needs per wave64
// instruction, with W waves per SIMD each running 8 independent chains of one instruction kind.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/ub/valu_rate tools/ub/valu_rate.hip      run: tools/ub/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
constexpr int kIters = 4096;
typedef float f2 __attribute__((ext_vector_type(2)));
template <int KIND> __global__ void __launch_bounds__(256) k_rate(float *out, float seed) {
	float a[8]; f2 p[8]; double d[8]; uint32_t u[8]; uint64_t q[8];
	for (int k = 0; k < 8; ++k) { a[k] = seed + k + threadIdx.x; p[k] = f2{a[k], a[k] + 1.f}; d[k] = a[k]; u[k] = (uint32_t)a[k] * 2654435761u; q[k] = (uint64_t)u[k] * 0x9E3779B97F4A7C15ull; }
	const float m = seed * 1.0000001f, c = seed * 0.5f;
	for (int it = 0; it < kIters; ++it) {
#define OP(K)                                                                                                             \
		if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[K]) : "v"(m), "v"(c));                             \
		if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[K]) : "v"(f2{m, m}), "v"(f2{c, c}));           \
		if (KIND == 2) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[K]) : "v"((double)m), "v"((double)c));            \
		if (KIND == 3) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(a[K]) : "v"(u[K]));                                   \
		if (KIND == 4) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[K]) : "v"(m), "v"(c));                           \
		if (KIND == 5) asm volatile("v_min_u32 %0, %0, %1" : "+v"(u[K]) : "v"(u[(K + 1) & 7]));                            \
		if (KIND == 6) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[K]) : "v"(m) : );                             \
		if (KIND == 7) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(a[K]) : "v"(1));                                       \
		if (KIND == 8) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[K]) : "v"(u[(K + 1) & 7]));                         \
		if (KIND == 9) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[K]));                                                      \
		if (KIND == 10) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[K]) : "v"((double)m));                                \
		if (KIND == 11) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[K]) : "v"((double)c));                                \
		if (KIND == 12) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a[K]) : "v"(u[K]), "v"(m)); \
		if (KIND == 13) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[K]) : "v"(u[(K + 1) & 7]));                            \
		if (KIND == 14) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[K]) : "v"(u[K]), "v"(u[(K + 1) & 7]) : "vcc"); \
		if (KIND == 15) asm volatile("v_rsq_f64 %0, %0" : "+v"(d[K]));                                                     \
		if (KIND == 16) asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0" : "+v"(d[K]) : "v"((double)m) : "vcc");         \
		if (KIND == 17) asm volatile("v_div_fmas_f64 %0, %0, %1, %2" : "+v"(d[K]) : "v"((double)m), "v"((double)c) : );   \
		if (KIND == 18) asm volatile("v_div_fixup_f64 %0, %0, %1, %2" : "+v"(d[K]) : "v"((double)m), "v"((double)c));     \
		if (KIND == 19) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d[K]) : "v"(u[K]));                                     \
		if (KIND == 20) asm volatile("v_lshrrev_b64 %0, 27, %0" : "+v"(q[K]));                                             \
		if (KIND == 21) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[K]) : "v"(d[K]));                                     \
		if (KIND == 22) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[K]) : "v"(u[(K + 1) & 7]));                        \
		if (KIND == 23) asm volatile("v_sqrt_f64 %0, %0" : "+v"(d[K]));
		REP8(OP)
#undef OP
	}
	float s = 0;
	for (int k = 0; k < 8; ++k) s += a[k] + p[k].x + p[k].y + (float)d[k] + (float)u[k] + (float)q[k];
	if (s == 12345.678f) out[0] = s;
}
template <int KIND> static void run(const char *name, float *out) {
	for (int waves = 1; waves <= 4; waves *= 2) { // waves per SIMD: blocks of 256 threads = 1 wave per SIMD of a CU
		const int blocks = 256 * waves;
		hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.5f);
		hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
		hipEventRecord(e0);
		hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.5f);
		hipEventRecord(e1); hipEventSynchronize(e1);
		float ms = 0; hipEventElapsedTime(&ms, e0, e1);
		const double instr_per_simd = (double)kIters * 8 * waves;
		printf("%-18s %d wave(s)/SIMD: %.3f ms -> %.2f ns per wave-instruction and SIMD = %.2f cycles at 2.4 GHz\n", name, waves, ms,
		       ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
	}
}
int main() {
	float *out = nullptr;
	if (hipMalloc(&out, 64) != hipSuccess) return 1;
	run<0>("v_fma_f32", out); run<1>("v_pk_fma_f32", out); run<2>("v_fma_f64", out); run<3>("v_cvt_f32_ubyte1", out);
	run<4>("v_max3_f32", out); run<5>("v_min_u32", out); run<6>("v_cndmask_b32", out); run<7>("v_ldexp_f32", out);
	run<8>("v_mul_lo_u32", out); run<9>("v_rcp_f64", out); run<10>("v_mul_f64", out); run<11>("v_add_f64", out);
	run<12>("v_fma_mix_f32", out); run<13>("v_add_u32", out);
	run<14>("v_mad_u64_u32", out); run<15>("v_rsq_f64", out); run<16>("v_div_scale_f64", out); run<17>("v_div_fmas_f64", out);
	run<18>("v_div_fixup_f64", out); run<19>("v_cvt_f64_u32", out); run<20>("v_lshrrev_b64", out); run<21>("v_cvt_f32_f64", out);
	run<22>("v_mul_hi_u32", out); run<23>("v_sqrt_f64", out);
	return hipDeviceSynchronize() == hipSuccess ? 0 : 2;
}
