// Device-side helpers shared by the radiosity translation units (rtmi_radiosity.hip: visibility, solve, display;
// rtmi_formfactor.hip: the form-factor build).  gfx950 only.
#pragma once
#include "rtmi_host.hpp"

#include <mutex>

struct rtmi_patches {
	rtmi_scene *scene;
	uint32_t n;
	double *d_verts;   // n × 9: v1, v2, v3 in ORIGINAL order (util.cpp:113-115 samples from these)
	double *d_normals; // n × 3: triNormal
	double *d_area;    // n: vec3.h:116-118
	uint32_t *d_noninv; // elements whose pseudo-inverse does not exist: inside for every point of their plane (util.cpp:146)
	uint32_t n_noninv;
	double *d_display; // n × 16 (display path): triNormal, triVert1, pseudo-inverse rows A, B (mat3x2.h:187-283), (−n)·v1, invertible, pad
	uint64_t verts_hash; // of the vertex array (must be the one the scene was built from)
	// tables of the form-factor kernel (rtmi_formfactor.hip), built on the first rtmi_form_factors_dev call
	std::mutex ff_mutex;
	bool ff_ready = false;
	int ff_kernel = 0; // which form-factor kernel serves this patch set: 0 = not measured yet, 1 = path walk, 2 = root traversal (ff_mutex)
	void *d_chain = nullptr;   // ChainEntry[]: per patch, the siblings along its root-to-leaf path
	void *d_meta = nullptr;    // PatchMeta[n]
	uint32_t *d_dfs = nullptr; // patch ids in depth-first leaf order, padded with RTMI_MISS to a multiple of 64
	void *d_gconst = nullptr;  // uint4[n_groups] {sure_margin, sure_cos (float bits), same-plane key, 0} + uint32[n] per-patch same-plane keys (rtmi_formfactor.hip)
	uint32_t n_groups = 0;     // 64-patch groups of d_dfs
	uint64_t chain_entries = 0;
	double plane_delta = -1.0; // tolerance of the "behind the patch's plane" tests, < 0: the tests are off
	double graze_cos_i = 1e-3, graze_cos_j = 1e-3; // rays leaving i / arriving on j with a smaller cosine test the flagged siblings of that path too (rtmi_formfactor.hip)
};

namespace rtmi {

constexpr double kEpsilon = 1.0e-5;       // Render2/base.h:57
// The reference's visibility test is Length(p − xj) < 1e-5 (radiosity.cpp:249).  The square root is correctly rounded and
// monotone, so sqrt(x) < 1e-5 holds exactly for the doubles x below the least x whose root rounds to >= 1e-5 — this constant
// (one ulp below fl(1e-5·1e-5); tests/test_host_abi.py checks both neighbours with the host's sqrt).  Comparing the squared
// length with it gives the same verdict bit for bit and saves a double-precision square root per ray.
constexpr double kEpsilonSquaredExact = 0x1.b7cdfd9d7bdbbp-34;
constexpr double kPi = 3.14159265358979;  // Render2/base.h:52
constexpr uint32_t kLightStream = 0xFFFFFFFFu;

struct rtmi_patches_dev_view { // what the display kernels read
	const double *verts, *normals, *display;
	uint32_t n;
	const uint32_t *noninv;
	uint32_t n_noninv;
};

struct PatchesDev {
	const double *verts, *normals, *area;
	uint32_t n;
};

// ---- counter-based sample generator (documented in DESIGN.md; the oracle restates it) ----------------
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {
	z += 0x9E3779B97F4A7C15ull;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}

// uniform point on patch e for stream (seed, a, b, k, which); barycentric mix as Render2/util.cpp:113-115
// (h0 = mix64(seed ^ (a << 32 | b)) is the same for every sample of a pair: callers may hoist it)
// ONE hash per point (round 3; two before): its upper 32 bits give u, its lower 32 bits v, each on a grid of 2^-32 — exact in
// double, one conversion instruction each instead of the four a 53-bit integer takes, and half the 64-bit multiplies (a
// 64-bit multiply is three 32-bit multiplies and an add3; they issue at the rate of other vector instructions on gfx950).  The reference draws both from an unseeded mt19937
// (util.cpp:96-124); the generator is ours; the CPU checker restates it, and
// tests/test_oracle_golden.py checks the stream's uniformity.
// bmin: the smallest of the point's three barycentric weights (exact: u, v and w = 1 − u − v are multiples of 2^-32)
__device__ __forceinline__ D3 sample_point_hb(const double *__restrict__ e, uint64_t h0, uint32_t k, uint32_t which, double &bmin) {
	const uint64_t h = mix64(h0 ^ (((uint64_t)k << 32) | which));
	uint32_t hi = (uint32_t)(h >> 32), lo = (uint32_t)h;
	// u = hi·2^-32, v = lo·2^-32.  The reflection `if (u + v > 1) { u = 1 − u; v = 1 − v; }` on the integers: u + v = (hi + lo)·2^-32
	// exactly (33 bits), so u + v > 1 <=> hi + lo > 2^32 <=> the 32-bit sum wraps and is not 0; 1 − u = (2^32 − hi)·2^-32 exactly,
	// and 2^32 − hi is the 32-bit 0 − hi (hi > 0 here).  The same u and v as the doubles give, bit for bit, two instructions fewer.
	const uint32_t sum = hi + lo;
	const bool reflect = (sum < hi) & (sum != 0u);
	hi = reflect ? 0u - hi : hi;
	lo = reflect ? 0u - lo : lo;
	// (seen as a 64-bit integer with a zero upper half, the conversion becomes cvt(0)·2^32 + cvt(hi): an addition of 0.0 the
	// compiler may not fold; behind this it is one v_cvt_f64_u32)
	asm volatile("" : "+v"(hi));
	const double u = (double)hi * 0x1.0p-32;
	const double v = (double)lo * 0x1.0p-32;
	const double w = 1.0 - u - v;
	// fmin() first canonicalises operands that come out of a select (v_max_f64 x, x — for signalling NaNs); none of the three
	// can be a NaN, so the bare instruction gives the same value
	double m;
	asm("v_min_f64 %0, %1, %2" : "=v"(m) : "v"(u), "v"(v));
	asm("v_min_f64 %0, %1, %2" : "=v"(bmin) : "v"(m), "v"(w));
	return d3((w * e[0]) + (u * e[3]) + (v * e[6]), (w * e[1]) + (u * e[4]) + (v * e[7]),
	          (w * e[2]) + (u * e[5]) + (v * e[8]));
}
__device__ __forceinline__ D3 sample_point_h(const double *__restrict__ e, uint64_t h0, uint32_t k, uint32_t which) {
	double unused;
	return sample_point_hb(e, h0, k, which, unused);
}
__device__ __forceinline__ D3 sample_point(const double *__restrict__ e, uint64_t seed, uint32_t a, uint32_t b,
                                           uint32_t k, uint32_t which) {
	return sample_point_h(e, mix64(seed ^ (((uint64_t)a << 32) | b)), k, which);
}

constexpr uint32_t kNormSlots = 64; // k_jacobi_sweep spreads its norm atomics over this many triples
// one Jacobi sweep (rtmi_radiosity.hip); block b adds its norms to triple b % slots of d_norms
// (ca, cb) != (1, 0): B_new = B_old + ca·(J(B_old) − B_old) + cb·(B_old − what d_B_new held), J(B) = E + rho·F·B — the Chebyshev step of solve_loop
int sweep_launch_slots(const float *d_F, uint64_t ld, uint32_t n, uint32_t row0, uint32_t n_rows, const double rho[3], const float *d_E,
                       const float *d_B_old, float *d_B_new, double *d_norms, uint32_t slots, void *stream, double ca = 1.0, double cb = 0.0);

// How the solve loop talks to the other ranks.  Two providers: the caller's rtmi_exchange_fn (rtmi_radiosity_solve_dev) and
// the library's own RCCL communicator (rtmi_comm.hip); a single-GPU solve passes none.
struct SolveExchange {
	void *self;
	// after every sweep, stream-ordered: make d_B_new hold every rank's rows.  A provider whose gather also reduces the sweep's
	// norm triple {Σd², Σ(B−E)², max|d|} across the ranks in place says so with triples_are_global.
	int (*gather)(void *self, float *d_B_new, double *d_triple, hipStream_t st);
	// at a check, stream-ordered: d_red = {Σd² last sweep, Σ(B−E)² last sweep, Σd² sweep before, max|d| last sweep} -> sums and
	// maximum over the ranks, in place (NULL when triples_are_global: nothing is left to reduce)
	int (*reduce)(void *self, double *d_red, hipStream_t st);
	bool triples_are_global;
};
// The solve loop behind rtmi_radiosity_solve_dev / _noalloc_dev / _sharded_dev (rtmi_radiosity.hip).  buf0 / buf1: the two
// ping-pong vectors of n_pad >= n rows (buf0 may be d_B itself when n_pad == n); d_norms: check_every windows of `slots`
// triples; d_red: 4 doubles; setup_rc != 0: this rank's set-up failed — the loop tells the other ranks before anyone enters
// a collective, and every rank returns -1.
int solve_loop(const char *who, const float *d_F, uint64_t ld, uint32_t n, uint32_t n_pad, uint32_t row0, uint32_t n_rows, const double rho[3],
               const float *d_E, float *d_B, float tol_rel, uint32_t max_sweeps, uint32_t check_every, const SolveExchange *X,
               rtmi_solve_stats *stats, hipStream_t run, float *buf0, float *buf1, double *d_norms, uint32_t slots, double *d_red,
               hipEvent_t e0, hipEvent_t e1, int setup_rc);

inline int check_patches(const rtmi_patches *p, const char *who) {
	if (!p) return fail(std::string(who) + ": patches is NULL");
	int dev = -1;
	if (hipGetDevice(&dev) != hipSuccess || dev != p->scene->device)
		return fail(std::string(who) + ": the patches live on another device than the current one");
	return 0;
}
inline PatchesDev pdev(const rtmi_patches *p) {
	PatchesDev d;
	d.verts = p->d_verts; d.normals = p->d_normals; d.area = p->d_area; d.n = p->n;
	return d;
}

// Render2/radiosity.cpp:231-262 (+ :202-229, object.cpp:23): is xj on patch j visible from xi?
// dir = Unit(xj − xi) and org = xi + 1e-5·dir are the caller's (the form-factor kernel reuses them).
__device__ __forceinline__ bool visible_along(const SceneDev &S, const PatchesDev &P, uint32_t j, D3 org, D3 dir, D3 yi, int *stack) {
	HitP h;
	trace<false, false, false>(S, org.x, org.y, org.z, dir.x, dir.y, dir.z, RTMI_INFINITY, stack, h); // binary nodes, see node_step
	if (h.pos == RTMI_MISS) return false;
	// hit.point = o + (d·((v1 − o)·n)) / (n·d)   with the HIT triangle's first vertex and normal
	const GTri &T = S.tris[h.pos];
	const double *np = P.normals + 3 * (size_t)T.orig;
	const D3 n = d3(np[0], np[1], np[2]);
	const D3 v1 = d3(T.v0[0], T.v0[1], T.v0[2]);
	const D3 p = org + (dot(v1 - org, n) * dir) / dot(n, dir);
	const double *nj = P.normals + 3 * (size_t)j;
	return len2(p - yi) < kEpsilonSquaredExact && dot(dir, d3(nj[0], nj[1], nj[2])) < 0; // sqrt(...) < 1e-5, see kEpsilonSquaredExact
}
__device__ __forceinline__ bool visible(const SceneDev &S, const PatchesDev &P, uint32_t j, D3 xi, D3 yi, int *stack) {
	const D3 dir = unit(yi - xi);
	return visible_along(S, P, j, xi + (kEpsilon * dir), dir, yi, stack);
}

} // namespace rtmi
