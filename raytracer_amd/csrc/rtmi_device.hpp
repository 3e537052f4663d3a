// Device-side data layout and the traversal core shared by the ray-query and radiosity kernels.
// gfx950 only.  Compile with -ffp-contract=off: the double-precision Möller–Trumbore below must
// round exactly like the reference's scalar C++ (Render2/object.cpp:29-60); fused multiply-adds are
// written out explicitly where they are wanted (fp32 box tests never need them).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rtmi.h"

namespace rtmi {

// ---------------------------------------------------------------- HBM layout
// BVH2 inner node, 32 B = a quarter of a 128-B L2 line = TWO 16-byte loads per lane: both children's boxes as
// 16-bit coordinates on a grid over the scene bounds (plane = qlo + q·qstep per axis; lo planes rounded down, hi
// planes up, so a quantised box contains the fp32 box) + the two child references.  The traversal kernels are
// bound by vector-memory instruction issue under divergent addresses (measured: every extra dependent load per
// node step costs as much as a quarter of the whole step), so halving the loads per node matters more than the
// ~24 extra VALU instructions of the dequantisation.
//   a = (lo0.x|hi0.x<<16, lo0.y|hi0.y<<16, lo0.z|hi0.z<<16, lo1.x|hi1.x<<16)
//   b = (lo1.y|hi1.y<<16, lo1.z|hi1.z<<16, ref0, ref1)
// ref >= 0: index of an inner node;  ref < 0: leaf, ~ref = (first << 3) | (count - 1), count <= 8,
// first = index into the leaf-ordered triangle array.
struct __attribute__((aligned(16))) QNode {
	uint4 a, b;
};

// 4-wide inner node, 48 B = THREE 16-byte loads per lane for up to four children (the binary node costs two loads per
// two children and twice the dependent steps; tools/est_bvh4.cpp: 149 instead of 182 requests per C3 ray).  The children's
// boxes are 8-bit coordinates on a grid local to the node: plane = origin + q·2^e per axis, origin = the node's own lo
// corner in fp32, e the smallest exponent with 255·2^e >= the node's extent; lo planes rounded down, hi planes up.
// Inner children come first and sit next to each other in the node array, leaf children (ONE triangle each) follow and
// their triangles sit next to each other in the triangle array, so two bases replace four references:
//   a = (origin.x, origin.y, origin.z [float bits], ex | ey<<8 | ez<<16 | n_inner<<24 | n_children<<28)   (e* = signed byte)
//   b = (lo.x of children 0..3 [one byte each], hi.x, lo.y, hi.y)
//   c = (lo.z, hi.z, index of inner child 0, ~(first << 3) with first = tri_base − n_inner: child k's leaf reference is c.w − 8k)
// Used by the ray queries; the display pass keeps the binary nodes (its line query has its own slab test).
struct __attribute__((aligned(16))) WNode {
	uint4 a, b, c;
};

// Triangle in leaf order, 80 B: v0 and the two edges the reference recomputes for every test
// (e1 = v1 − v0, e2 = v2 − v0: the same IEEE subtraction done once on the host), original id, object.
struct __attribute__((aligned(16))) GTri {
	double v0[3], e1[3], e2[3];
	uint32_t orig, object;
};

constexpr int kStackDepth = 32;   // cap on the depth of the binary hierarchy (builder); the LDS stack of a launch is sized per scene
constexpr int kBlock = 256;       // threads per workgroup in every traversal kernel
constexpr int kSentinel = INT32_MIN;      // lane state: no node left (ray finished, or lane idle)
constexpr uint32_t kGuard = 1u << 26;      // default of SceneDev::guard: wave-iterations per ray after which a traversal gives up (safety net)
#ifndef RTMI_NODE_STEPS
#define RTMI_NODE_STEPS 4
#endif
constexpr int kNodeSteps = RTMI_NODE_STEPS;  // node steps a wave takes per scheduling round of trace_queue
constexpr uint32_t kQueueChunk = 256;     // items a wave takes from the launch's queue at a time (4 squares of 8x8 pixels) ...
#ifndef RTMI_TAIL_CHUNK
#define RTMI_TAIL_CHUNK 128
#endif
constexpr uint32_t kTailChunk = RTMI_TAIL_CHUNK; // ... and once the queue is down to its last chunk per resident wave (see trace_queue)
constexpr int kLeaf = INT32_MIN + 1;      // lane state: working through the triangles of a leaf
#ifndef RTMI_SPLIT
#define RTMI_SPLIT 0
#endif
#ifndef RTMI_POOL
#define RTMI_POOL 1
#endif
#ifndef RTMI_TAIL_SHIFT
#define RTMI_TAIL_SHIFT 2
#endif
constexpr uint32_t kTailShift = RTMI_TAIL_SHIFT; // the last claims of a launch are no smaller than the paced size >> this
#ifndef RTMI_POOL_TAKE
#define RTMI_POOL_TAKE 64
#endif
constexpr uint32_t kPoolTake = RTMI_POOL_TAKE;        // items a wave moves from the shared range into its own at a time
constexpr bool kPool = RTMI_POOL != 0;    // the waves of a workgroup share what they claim from the launch's queue (trace_queue)
constexpr bool kSplit = RTMI_SPLIT != 0;  // closest-hit queues: once a wave has no items left, its idle lanes take subtrees off the busy lanes' stacks (trace_queue)
#ifndef RTMI_SPLIT_PAIRS
#define RTMI_SPLIT_PAIRS 16
#endif
constexpr int kSplitPairs = RTMI_SPLIT_PAIRS; // subtrees handed over per scheduling round, at most
#ifndef RTMI_TOP_LDS
#define RTMI_TOP_LDS 0
#endif
constexpr uint32_t kTopNodes = 21;        // root + 4 + 16: the levels trace_queue can keep in LDS (RTMI_TOP_LDS=1; measured)
// Traversal kernels are latency-bound: ask the register allocator for at least this many waves per SIMD.
#ifndef RTMI_WAVES
#define RTMI_WAVES 5
#endif
#define RTMI_TRAVERSAL_KERNEL_W(w) __global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(w, 8)))
#define RTMI_TRAVERSAL_KERNEL RTMI_TRAVERSAL_KERNEL_W(RTMI_WAVES)
constexpr uint32_t kMaxFirst = 1u << 27; // leaf refs ~((first << 3) | cnt-1) stay above kLeaf

struct SceneDev {
	const QNode *nodes;   // binary, display pass
	const WNode *wnodes;  // 4-wide, ray queries; root = 0 in both when root_ref >= 0
	float qlo[3], qstep[3]; // the quantisation grid: plane = qlo + q·qstep (q = 0..65535)
	const GTri *tris;
	int32_t root_ref; // kSentinel for an empty scene
	float absmax;     // max |coordinate| of the scene bounds (for the per-ray box padding)
	uint32_t n_top;   // wnodes[0 .. n_top) = the top three levels of the 4-wide hierarchy (<= kTopNodes): staged in LDS by trace_queue
	uint32_t guard;   // wave-iterations a traversal may spend on one ray (kGuard; RTMI_GUARD overrides it, tests)
	int *error;       // host-mapped word, set to 1 when a wave gives up: the host entries report it (rtmi_scene_check)
};

// A wave that cannot finish must still end — and must not look like success: it raises the scene's error word.
__device__ __forceinline__ void give_up(const SceneDev &S) { __hip_atomic_store(S.error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// ---------------------------------------------------------------- exact triangle test
// Render2/object.cpp:29-60 in double, same operation order; inclusive edges; 0.0 = miss.
// u, v are only defined up to the point the reference assigns them.
__device__ __forceinline__ double mt_exact(double ox, double oy, double oz, double dx, double dy, double dz,
                                           const GTri &T, double &u, double &v) {
	const double e1x = T.e1[0], e1y = T.e1[1], e1z = T.e1[2];
	const double e2x = T.e2[0], e2y = T.e2[1], e2z = T.e2[2];
	// pvec = dir ^ e2
	const double px = dy * e2z - dz * e2y;
	const double py = dz * e2x - dx * e2z;
	const double pz = dx * e2y - dy * e2x;
	const double det = (e1x * px) + (e1y * py) + (e1z * pz);
	if (det < 1e-8 && det > -1e-8) return 0.0;
	const double inv_det = 1.0 / det;
	const double tx = ox - T.v0[0], ty = oy - T.v0[1], tz = oz - T.v0[2];
	u = ((tx * px) + (ty * py) + (tz * pz)) * inv_det;
	if (u < 0.0 || u > 1.0) return 0.0;
	// qvec = tvec ^ e1
	const double qx = ty * e1z - tz * e1y;
	const double qy = tz * e1x - tx * e1z;
	const double qz = tx * e1y - ty * e1x;
	v = ((dx * qx) + (dy * qy) + (dz * qz)) * inv_det;
	if (v < 0.0 || u + v > 1.0) return 0.0;
	return ((e2x * qx) + (e2y * qy) + (e2z * qz)) * inv_det;
}

// A triangle record in exactly FIVE aligned 16-byte loads issued together.  (Reading the fields one by one, the compiler
// fetched overlapping pieces — six loads — and sank those of v0 behind the det window test: a second, dependent memory
// round trip in every triangle step.)
__device__ __forceinline__ GTri load_tri(const GTri *p) {
	typedef double d2 __attribute__((ext_vector_type(2)));
	const d2 *q = reinterpret_cast<const d2 *>(p);
	const d2 a = q[0], b = q[1], c = q[2], d = q[3], e = q[4];
	asm volatile("" : : "v"(a), "v"(b), "v"(c), "v"(d), "v"(e));
	GTri T;
	T.v0[0] = a.x; T.v0[1] = a.y; T.v0[2] = b.x;
	T.e1[0] = b.y; T.e1[1] = c.x; T.e1[2] = c.y;
	T.e2[0] = d.x; T.e2[1] = d.y; T.e2[2] = e.x;
	const unsigned long long w = (unsigned long long)__double_as_longlong(e.y);
	T.orig = (uint32_t)w;
	T.object = (uint32_t)(w >> 32);
	return T;
}

// ---------------------------------------------------------------- conservative fp32 slab test
// The boxes only steer; the exact test decides.  They must never cull a triangle the exact test
// would accept, so every slab is widened per ray by pad = 2^-18·(scene |coord| max + |o|max), which
// dominates (a) the double->float rounding of o and d, (b) the <=5 fp32 roundings below and (c) the
// distance by which a ray accepted by the double MT can miss the triangle's true bounds
// (DESIGN.md §"Why the boxes are safe").  The pad is folded into two copies of the origin, so it
// costs no instruction per box.
// Arithmetic: plane = qlo + q·qstep, t = (plane − (o ± pad))/d = q·(qstep/d) + (qlo − (o ± pad))/d: ONE fused
// multiply-add per plane on the integer coordinate.  The pre-rounded constants carry errors of a few 2^-24·|coord|
// in distance units, far below the pad.
struct RayF {
	float ax, ay, az;    // qstep/d, |d| clamped away from 0 (no inf, hence no NaN from 0·inf)
	float blx, bly, blz; // (qlo − (o + pad))/d : addend of the lo planes
	float bhx, bhy, bhz; // (qlo − (o − pad))/d : addend of the hi planes
};

__device__ __forceinline__ float safe_rcp(float d) {
	const float tiny = 1e-20f;
	if (fabsf(d) < tiny) d = (d < 0.f) ? -tiny : tiny;
	// (v_rcp_f32 — 1 ulp, safe under the pad — instead of the ten-instruction IEEE division was measured: no difference, -DRTMI_FAST_RCP)
#ifdef RTMI_FAST_RCP
	return __builtin_amdgcn_rcpf(d);
#else
	return 1.0f / d;
#endif
}

__device__ __forceinline__ RayF make_rayf(const SceneDev &S, double ox, double oy, double oz, double dx, double dy, double dz) {
	const float fx = (float)ox, fy = (float)oy, fz = (float)oz;
	const float pad = 3.814697265625e-06f * (S.absmax + fmaxf(fabsf(fx), fmaxf(fabsf(fy), fabsf(fz)))) + 1e-30f;
	const float ix = safe_rcp((float)dx), iy = safe_rcp((float)dy), iz = safe_rcp((float)dz);
	RayF r;
	r.ax = S.qstep[0] * ix; r.ay = S.qstep[1] * iy; r.az = S.qstep[2] * iz;
	r.blx = fmaf(S.qlo[0], ix, -(fx + pad) * ix); r.bly = fmaf(S.qlo[1], iy, -(fy + pad) * iy); r.blz = fmaf(S.qlo[2], iz, -(fz + pad) * iz);
	r.bhx = fmaf(S.qlo[0], ix, -(fx - pad) * ix); r.bhy = fmaf(S.qlo[1], iy, -(fy - pad) * iy); r.bhz = fmaf(S.qlo[2], iz, -(fz - pad) * iz);
	return r;
}

// one box: the three (lo | hi << 16) words of its planes
__device__ __forceinline__ bool slab(uint32_t wx, uint32_t wy, uint32_t wz, const RayF &r, float tbest, float &tnear) {
	const float ax = fmaf((float)(wx & 0xffffu), r.ax, r.blx), bx = fmaf((float)(wx >> 16), r.ax, r.bhx);
	const float ay = fmaf((float)(wy & 0xffffu), r.ay, r.bly), by = fmaf((float)(wy >> 16), r.ay, r.bhy);
	const float az = fmaf((float)(wz & 0xffffu), r.az, r.blz), bz = fmaf((float)(wz >> 16), r.az, r.bhz);
	const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
	const float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), tbest));
	tnear = tn;
	return tn <= tf;
}

// One BINARY node step of a lane: fetch the node (2 × 16 B), test both children, go to the nearer hit child and push the
// other one, or pop.  Returns the lane's next reference (inner node, raw leaf reference, or kSentinel).  Kept beside the
// 4-wide step below for the form-factor kernel, whose lanes all start together and never refill: it is bound by
// vector-ALU issue at decaying lane occupancy, and two cheap steps cost it less than one wide one (measured on the
// 64 K-patch box: 8.7 against 8.2 G pair-samples/s).
__device__ __forceinline__ int node_step(const SceneDev &S, const RayF &r, float tbest, int cur, int &sp, int *stack) {
	const QNode *np = S.nodes + cur;
	const uint4 a = np->a, b = np->b;
	float t0, t1;
	const bool h0 = slab(a.x, a.y, a.z, r, tbest, t0);
	const bool h1 = slab(a.w, b.x, b.y, r, tbest, t1);
	const int r0 = (int)b.z, r1 = (int)b.w;
	if (h0 && h1) {
		const bool swap = t1 < t0;
		stack[sp * kBlock] = swap ? r0 : r1;
		++sp;
		return swap ? r1 : r0;
	}
	if (h0) return r0;
	if (h1) return r1;
	if (sp > 0) {
		--sp;
		return stack[sp * kBlock];
	}
	return kSentinel;
}

// ---------------------------------------------------------------- the 4-wide node step of the ray queries
// t(plane) = (origin + q·2^e − (o ± pad))/d = q·(2^e/d) + (origin/d − (o ± pad)/d): 2^e/d is exact (v_ldexp_f32), the
// addend costs one fused multiply-add per axis and side and node on the pre-rounded −(o ± pad)/d, then ONE fused
// multiply-add per plane on the byte (v_cvt_f32_ubyteN).  Which plane of an axis the ray enters through is known from
// the sign of d, so the node's lo/hi words are swapped once per node and no per-child min/max pair is needed.
// Rounding errors: a few 2^-24·|coord| in distance units, far below the pad (as above).
struct RayW {
	float ix, iy, iz;    // 1/d, |d| clamped away from 0
	float cnx, cny, cnz; // addend of the entry planes: −(o + pad)/d for d >= 0 (lo plane), −(o − pad)/d for d < 0 (hi plane)
	float cfx, cfy, cfz; // addend of the exit planes: the other one
};

__device__ __forceinline__ RayW make_rayw(const SceneDev &S, double ox, double oy, double oz, double dx, double dy, double dz) {
	const float fx = (float)ox, fy = (float)oy, fz = (float)oz;
	const float pad = 3.814697265625e-06f * (S.absmax + fmaxf(fabsf(fx), fmaxf(fabsf(fy), fabsf(fz)))) + 1e-30f;
	RayW r;
	r.ix = safe_rcp((float)dx); r.iy = safe_rcp((float)dy); r.iz = safe_rcp((float)dz);
	const float lx = -(fx + pad) * r.ix, ly = -(fy + pad) * r.iy, lz = -(fz + pad) * r.iz;
	const float hx = -(fx - pad) * r.ix, hy = -(fy - pad) * r.iy, hz = -(fz - pad) * r.iz;
	r.cnx = r.ix < 0.f ? hx : lx; r.cny = r.iy < 0.f ? hy : ly; r.cnz = r.iz < 0.f ? hz : lz;
	r.cfx = r.ix < 0.f ? lx : hx; r.cfy = r.iy < 0.f ? ly : hy; r.cfz = r.iz < 0.f ? lz : hz;
	return r;
}

template <int K> __device__ __forceinline__ float ubyte(uint32_t w) { return (float)((w >> (8 * K)) & 0xffu); }
__device__ __forceinline__ int sbyte(uint32_t w, int k) { return (int)(w << (24 - 8 * k)) >> 24; }

// One node step of a lane: fetch the node (3 × 16 B), test its children, go to the nearest hit child and push the others
// (farthest first), or pop.  Returns the lane's next reference (inner node, raw leaf reference, or kSentinel).
// A hit child is carried as ONE sortable word: the bits of its entry distance (>= 0, so they order like the float) with
// the child's number in the two lowest bits; 0xffffffff = not hit.  Five min/max pairs sort the four words.
#ifndef RTMI_PAIR_FETCH
#define RTMI_PAIR_FETCH 0
#endif
// RTMI_PAIR_FETCH=1 (experiment, profiles/ab_r03.md): neighbouring lanes 2k, 2k+1 fetch their two nodes together.  Every lane
// loads piece a of its own node; then the pair loads pieces b and c of the even lane's node with ONE instruction (the two
// addresses are neighbours in one cache line seven times in eight) and b and c of the odd lane's node with another; a
// quad_perm swap hands each lane the piece its neighbour fetched for it.  Four cache-line look-ups per pair and visit
// instead of six, for about twenty more vector instructions.  Must be called by every lane of the wave.
__device__ __forceinline__ int dpp_swap_neighbour(int x) { return __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true); } // quad_perm [1, 0, 3, 2]
__device__ __forceinline__ uint4 dpp_swap_neighbour(uint4 v) {
	return make_uint4((uint32_t)dpp_swap_neighbour((int)v.x), (uint32_t)dpp_swap_neighbour((int)v.y), (uint32_t)dpp_swap_neighbour((int)v.z),
	                  (uint32_t)dpp_swap_neighbour((int)v.w));
}
__device__ __forceinline__ void pair_fetch(const SceneDev &S, int cur, uint32_t lane, uint4 &a, uint4 &b, uint4 &c) {
	const bool odd = (lane & 1u) != 0u;
	const int pcur = dpp_swap_neighbour(cur);
	const int n0 = odd ? pcur : cur, n1 = odd ? cur : pcur; // the even lane's node, the odd lane's node
	const uint4 *base = reinterpret_cast<const uint4 *>(S.wnodes);
	uint4 r1 = make_uint4(0, 0, 0, 0), r2 = r1, r3 = r1;
	if (cur >= 0) r1 = base[3 * (size_t)(uint32_t)cur];                       // a of my own node
	if (n0 >= 0) r2 = base[3 * (size_t)(uint32_t)n0 + (odd ? 2u : 1u)];      // even: b(n0), odd: c(n0)
	if (n1 >= 0) r3 = base[3 * (size_t)(uint32_t)n1 + (odd ? 2u : 1u)];      // even: b(n1), odd: c(n1)
	const uint4 x = dpp_swap_neighbour(r2), y = dpp_swap_neighbour(r3);
	a = r1;
	b = odd ? y : r2;  // odd: the b(n1) its neighbour fetched
	c = odd ? r3 : x;  // even: the c(n0) its neighbour fetched
}

__device__ __forceinline__ int node_step_abc(const SceneDev &S, const RayW &r, float tbest, uint4 a, uint4 b, uint4 c, int &sp, int *stack, uint32_t allow = 15u);
__device__ __forceinline__ int node_step(const SceneDev &S, const RayW &r, float tbest, int cur, int &sp, int *stack, const WNode *top = nullptr,
                                         uint32_t n_top = 0) {
#ifdef RTMI_NODE_STRIDE // experiment: nodes at a stride of 64 B (none straddles a cache line), 48 B of each used
	const WNode *np = reinterpret_cast<const WNode *>(reinterpret_cast<const char *>(S.wnodes) + (size_t)(uint32_t)cur * RTMI_NODE_STRIDE);
#else
	const WNode *np = S.wnodes + cur;
#endif
	uint4 a, b, c;
#if RTMI_TOP_LDS
	if ((uint32_t)cur < n_top) { a = top[cur].a; b = top[cur].b; c = top[cur].c; } // the levels every ray starts in: from LDS
	else
#endif
	{ a = np->a; b = np->b; c = np->c; }
	return node_step_abc(S, r, tbest, a, b, c, sp, stack);
}
// Experiment (RTMI_DRAIN_PREFETCH, profiles/ab_r03.md): once the launch's queue is dry a wave's time is its rays' chains of dependent
// node fetches (≈ 2 µs per round of three steps, most of it waiting for L2).  A node's inner children lie next to each other
// (c.z + k, 48 B each), so as soon as a node's words are in, the step asks for their lines with one or two loads nobody waits for;
// the slab arithmetic runs meanwhile and the next step's own fetch finds the lines in L1.  gfx950 has no vector prefetch
// instruction, and an ordinary load into a register nobody waits for may land after the compiler has given that register to
// something else; so these are loads that write to LDS (global_load_lds_dword: 64 dwords at M0, a landing strip per wave that
// nothing reads) — no register is involved, and the compiler's own waits only become more conservative.
#ifndef RTMI_DRAIN_PREFETCH
#define RTMI_DRAIN_PREFETCH 0
#endif
__device__ __forceinline__ int node_step_prefetching(const SceneDev &S, const RayW &r, float tbest, int cur, int &sp, int *stack, bool dry, uint32_t land) {
	const WNode *np = S.wnodes + cur;
	const uint4 a = np->a, b = np->b, c = np->c;
	if (dry) { // wave-uniform
		const uint32_t n_inner = (a.w >> 24) & 0xfu;
		const char *kids = reinterpret_cast<const char *>(S.wnodes + c.z);
		// (no instruction offset: it would move the landing address in LDS as well — the strip of the workgroup's last wave then
		// reached into the traversal stacks behind it, and a launch faulted)
		if (n_inner > 0u) asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dword %0, off" : : "v"(kids), "s"(land) : "m0", "memory");
		if (n_inner > 2u) asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dword %0, off" : : "v"(kids + 128), "s"(land) : "m0", "memory"); // (bytes 128.. exist: three children are 144)
	}
	return node_step_abc(S, r, tbest, a, b, c, sp, stack);
}

// allow: which of the node's children take part (bit K = child K; 15 = all, folded away by the compiler) — a walk that starts at a
// binary node the collapse opened enters its wide node for SOME of the children only (Wide4::cover, node_step_cover)
__device__ __forceinline__ int node_step_abc(const SceneDev &S, const RayW &r, float tbest, uint4 a, uint4 b, uint4 c, int &sp, int *stack, uint32_t allow) {
	const float ax = ldexpf(r.ix, sbyte(a.w, 0)), ay = ldexpf(r.iy, sbyte(a.w, 1)), az = ldexpf(r.iz, sbyte(a.w, 2));
	const float px = __uint_as_float(a.x), py = __uint_as_float(a.y), pz = __uint_as_float(a.z);
	const float bnx = fmaf(px, r.ix, r.cnx), bny = fmaf(py, r.iy, r.cny), bnz = fmaf(pz, r.iz, r.cnz);
	const float bfx = fmaf(px, r.ix, r.cfx), bfy = fmaf(py, r.iy, r.cfy), bfz = fmaf(pz, r.iz, r.cfz);
	const bool sx = r.ix < 0.f, sy = r.iy < 0.f, sz = r.iz < 0.f;
	const uint32_t nxw = sx ? b.y : b.x, fxw = sx ? b.x : b.y;
	const uint32_t nyw = sy ? b.w : b.z, fyw = sy ? b.z : b.w;
	const uint32_t nzw = sz ? c.y : c.x, fzw = sz ? c.x : c.y;
	const uint32_t n_inner = (a.w >> 24) & 0xfu, n_child = a.w >> 28;
	uint32_t key[4];
#define RTMI_CHILD(K)                                                                                                         \
	{                                                                                                                         \
		const float tn = fmaxf(fmaxf(fmaf(ubyte<K>(nxw), ax, bnx), fmaf(ubyte<K>(nyw), ay, bny)), fmaxf(fmaf(ubyte<K>(nzw), az, bnz), 0.0f)); \
		const float tf = fminf(fminf(fmaf(ubyte<K>(fxw), ax, bfx), fmaf(ubyte<K>(fyw), ay, bfy)), fminf(fmaf(ubyte<K>(fzw), az, bfz), tbest)); \
		key[K] = (tn <= tf && (K < 2 || n_child > K) && ((allow >> K) & 1u)) ? ((__float_as_uint(tn) & ~3u) | K) : 0xffffffffu; \
	}
	RTMI_CHILD(0) RTMI_CHILD(1) RTMI_CHILD(2) RTMI_CHILD(3) // a node has at least two children (check_wide): no count test for 0 and 1
#undef RTMI_CHILD
#define RTMI_CE(I, J)                                                                                                         \
	{                                                                                                                         \
		const uint32_t lo = min(key[I], key[J]), hi = max(key[I], key[J]);                                                    \
		key[I] = lo; key[J] = hi;                                                                                             \
	}
	RTMI_CE(0, 1) RTMI_CE(2, 3) RTMI_CE(0, 2) RTMI_CE(1, 3) RTMI_CE(1, 2)
#undef RTMI_CE
	// child k: inner -> c.z + k;  leaf -> ~((first + k) << 3) = c.w − 8k with c.w = ~(first << 3) (one triangle: count − 1 = 0)
	auto ref_of = [&](uint32_t k) -> int { return k < n_inner ? (int)(c.z + k) : (int)(c.w - (k << 3)); };
	if (key[0] == 0xffffffffu) {
		if (sp > 0) {
			--sp;
			return stack[sp * kBlock];
		}
		return kSentinel;
	}
	if (key[3] != 0xffffffffu) { stack[sp * kBlock] = ref_of(key[3] & 3u); ++sp; }
	if (key[2] != 0xffffffffu) { stack[sp * kBlock] = ref_of(key[2] & 3u); ++sp; }
	if (key[1] != 0xffffffffu) { stack[sp * kBlock] = ref_of(key[1] & 3u); ++sp; }
	return ref_of(key[0] & 3u);
}

// A node step from a reference that may name only some of a node's children: node index in bits 0..26, children in bits 27..30
// (0 = all) — what Wide4::cover holds and what this step's own pushes are (plain indices).
__device__ __forceinline__ int node_step_cover(const SceneDev &S, const RayW &r, float tbest, int cur, int &sp, int *stack) {
	const uint32_t some = ((uint32_t)cur >> 27) & 15u;
	const WNode *np = S.wnodes + ((uint32_t)cur & 0x7ffffffu);
	const uint4 a = np->a, b = np->b, c = np->c;
	return node_step_abc(S, r, tbest, a, b, c, sp, stack, some ? some : 15u);
}

// float >= x (x >= 0)
__device__ __forceinline__ float f32_above(double x) { return __double2float_ru(x); }

// ---------------------------------------------------------------- traversal
// One ray per lane, per-lane stack in LDS (`stack` points at this lane's column, entries are
// kBlock ints apart so a wave's pushes and pops hit 64 different banks).
// ANY = false: closest hit — strict minimum of the positive t below tmax; equal t -> smaller
//              original id (the in-order scan of Render2/object.cpp:143-168 keeps the first).
// ANY = true : stop at the first triangle with 0 < t < tmax (Render2/shader.cpp:131,226).
// COUNT = true additionally tallies node visits and triangle tests (the roofline's work count).
struct Work { uint32_t nodes, tests, node_slots, test_slots, wait_leaf, wait_idle; }; // *_slots: 64 per wave-iteration (lane occupancy = nodes / node_slots)
__device__ __forceinline__ uint32_t wave_slots() { // 64 on the first active lane of the wave, 0 elsewhere
	const unsigned long long m = __ballot(1);
	return (uint32_t)(__ffsll((long long)m) - 1) == (threadIdx.x & 63u) ? 64u : 0u;
}

__device__ __forceinline__ double readlane_f64(double x, int k) { // k wave-uniform
	const unsigned long long b = (unsigned long long)__double_as_longlong(x);
	const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, k), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), k);
	return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

__device__ __forceinline__ double shfl_f64(double x, uint32_t src) { // every lane of the wave must be active
	const unsigned long long b = (unsigned long long)__double_as_longlong(x);
	const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)b, (int)src, 64), hi = (uint32_t)__shfl((int)(uint32_t)(b >> 32), (int)src, 64);
	return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// position of the n-th set bit of m (n < popcount(m)); m wave-uniform, n per lane
__device__ __forceinline__ uint32_t nth_set_bit64(unsigned long long m, uint32_t n) {
	uint32_t w = (uint32_t)m, base = 0;
	const uint32_t c = (uint32_t)__popc(w);
	if (n >= c) { n -= c; w = (uint32_t)(m >> 32); base = 32; }
#pragma unroll
	for (uint32_t s = 16; s >= 1; s >>= 1) {
		const uint32_t c2 = (uint32_t)__popc(w & ((1u << s) - 1u));
		if (n >= c2) { n -= c2; w >>= s; base += s; }
	}
	return base;
}

#ifdef RTMI_TAIL_DIAG
static __device__ unsigned long long g_tail_diag[5 * 8192]; // per wave of the last primary-ray launch (this translation unit's copy)
#endif

struct HitP { double t; uint32_t pos; }; // pos: index into the leaf-ordered triangle array, RTMI_MISS = no hit (t = 0)

template <bool WIDE> struct StepRay;
template <> struct StepRay<true> {
	typedef RayW type;
	static __device__ __forceinline__ RayW make(const SceneDev &S, double ox, double oy, double oz, double dx, double dy, double dz) { return make_rayw(S, ox, oy, oz, dx, dy, dz); }
};
template <> struct StepRay<false> {
	typedef RayF type;
	static __device__ __forceinline__ RayF make(const SceneDev &S, double ox, double oy, double oz, double dx, double dy, double dz) { return make_rayf(S, ox, oy, oz, dx, dy, dz); }
};

// WIDE = false walks the binary nodes (stack: rtmi_scene::stack_bytes_binary suffices), true the 4-wide ones.
template <bool ANY, bool COUNT = false, bool WIDE = true>
__device__ __forceinline__ void trace(const SceneDev &S, double ox, double oy, double oz, double dx, double dy,
                                      double dz, double tmax, int *stack, HitP &h, Work *w = nullptr) {
	h.t = 0.0;
	h.pos = RTMI_MISS;
	double best = tmax;
	float tbest = f32_above(best);
	const typename StepRay<WIDE>::type r = StepRay<WIDE>::make(S, ox, oy, oz, dx, dy, dz);
	int sp = 0;
	int cur = S.root_ref;
	uint32_t tk = 0, tend = 0;
	// All lanes start together and the wave runs until its last ray is done; per iteration it takes ONE node
	// step or ONE triangle step, whichever more lanes wait for (see trace_queue for the refilling variant).
	for (uint32_t guard = 0;; ++guard) {
		if (cur < 0 && cur != kSentinel && cur != kLeaf) {
			const uint32_t code = ~(uint32_t)cur;
			tk = code >> 3;
			tend = tk + (code & 7u) + 1u;
			cur = kLeaf;
		}
		const int n_node = __popcll(__ballot(cur >= 0)), n_leaf = __popcll(__ballot(cur == kLeaf));
		if (n_node + n_leaf == 0) break;
		if (guard >= S.guard) { give_up(S); h.t = 0.0; h.pos = RTMI_MISS; break; } // never reached on valid input
		if (n_node >= n_leaf) {
			if (cur >= 0) {
				if (COUNT) { ++w->nodes; w->node_slots += wave_slots(); }
				cur = node_step(S, r, tbest, cur, sp, stack);
			}
		} else if (cur == kLeaf) {
			if (COUNT) { ++w->tests; w->test_slots += wave_slots(); }
			const GTri T = load_tri(S.tris + tk);
			double u = 0.0, v = 0.0;
			const double t = mt_exact(ox, oy, oz, dx, dy, dz, T, u, v);
			bool done = false;
			if (ANY) {
				if (t > 0.0 && t < tmax) {
					h.t = t; h.pos = tk;
					done = true;
				}
			} else {
				if (t > 0.0 && (best > t || (t == best && h.pos != RTMI_MISS && T.orig < S.tris[h.pos].orig))) {
					best = t;
					tbest = f32_above(best);
					h.t = t; h.pos = tk;
				}
			}
			++tk;
			if (done || (tk == tend && sp == 0)) {
				cur = kSentinel;
			} else if (tk == tend) {
				--sp;
				cur = stack[sp * kBlock];
			}
		}
	}
}

// ---------------------------------------------------------------- persistent waves over a work queue
// refill threshold and phase-vote weights of trace_queue (RTMI_REFILL / RTMI_VOTE_NODE / RTMI_VOTE_LEAF override)
struct Tune { int thresh, vote_node, vote_leaf; uint32_t chunk, tail_chunk, max_chunk; }; // chunk: items per claim (kQueueChunk unless the launch is small)

// The launch owns `n_items` work items (rays) and ONE counter in device memory (zeroed on the stream before the
// kernel); its grid is persistent — as many workgroups as the device holds at once.  Every lane traces one ray at
// a time; whenever at least `thresh` lanes of a wave are idle, the wave claims that many items with ONE atomic
// (ballot + mbcnt give each idle lane its slot) and the idle lanes start new rays while the others carry on.
// (A counter per workgroup in LDS over 2048 items was the first form: its waves idled at the end of every
// workgroup's queue — 30 % of the lanes of a node step had no ray.)  Ray lengths in a scene are close to exponentially distributed, so without this a wave runs for
// the longest of its 64 rays (measured 20 % lane occupancy in the node loop on the 1 Mi-triangle soup).
//   fetch(item, R) -> bool    fill the first ray of a work item (false: no ray for this item)
//   emit(item, h, R) -> bool  consume the hit record; true = R now holds the item's NEXT ray (form factors:
//                             S samples per matrix entry, summed in sample order by the lane that owns it)
struct RayD { double ox, oy, oz, dx, dy, dz, tmax; };
// What a lane keeps of its best hit while traversing: the triangle's position in the leaf-ordered array
// (original id and object are read back from there when the ray retires), t (in `best`), and u, v already
// rounded to the float the hit record stores.  Few registers = more waves per SIMD.
struct HitQ { double t; float u, v; uint32_t pos; }; // pos == RTMI_MISS: no hit (then t = 0)

// UNIFORM_O: every ray of the launch starts at (uox, uoy, uoz) (primary rays: the eye) — the origin then
// lives in scalar registers instead of six VGPRs per lane; fetch() need not fill R.ox..oz.
template <bool ANY, bool COUNT, bool UNIFORM_O, class Fetch, class Emit>
__device__ __forceinline__ void trace_queue(const SceneDev &S, int *stack, int *counter, uint32_t n_items, const Tune &tune,
                                            double uox, double uoy, double uoz, Fetch fetch, Emit emit, Work *w = nullptr) {
	const int thresh = tune.thresh, vote_node = tune.vote_node, vote_leaf = tune.vote_leaf;
	const uint32_t lane = threadIdx.x & 63u;
#if RTMI_TOP_LDS
	__shared__ WNode top[kTopNodes];
	const uint32_t n_top = min(S.n_top, kTopNodes);
	if (threadIdx.x < 3u * n_top) reinterpret_cast<uint4 *>(top)[threadIdx.x] = reinterpret_cast<const uint4 *>(S.wnodes)[threadIdx.x];
	__syncthreads();
#else
	const WNode *top = nullptr;
	const uint32_t n_top = 0;
	(void)top; (void)n_top;
#endif
	__shared__ unsigned long long pool_word;
	__shared__ uint32_t pool_seen_word;
	unsigned long long *pool = &pool_word;
	__shared__ uint32_t pool_sum_word;        // items the workgroup claimed after its first claim
	__shared__ unsigned long long pool_t_word; // when it first claimed from the launch's counter (wall_clock64, 100 MHz)
	volatile uint32_t *pool_seen = &pool_seen_word, *pool_sum = &pool_sum_word;
	volatile unsigned long long *pool_t = &pool_t_word;
	if (kPool) {
		if (threadIdx.x == 0) { pool_word = 0ull; pool_seen_word = 0u; pool_sum_word = 0u; pool_t_word = 0ull; }
		__syncthreads();
	}
	RayD R = {0, 0, 0, 0, 0, 1, 0};
	RayW r = {0, 0, 0, 0, 0, 0, 0, 0, 0};
	HitQ h = {0.0, 0.f, 0.f, RTMI_MISS};
	double best = 0.0;
	float tbest = 0.f;
	int sp = 0, cur = kSentinel;
	uint32_t item = 0, tk = 0, tend = 0; // [tk, tend): triangles of the leaf this lane is working through (cur == kLeaf)
#if RTMI_DRAIN_PREFETCH
	__shared__ uint32_t pf_strip[kBlock]; // the landing strip of the drain's prefetch loads: 64 dwords per wave, never read
	const uint32_t pf_land = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(pf_strip + (threadIdx.x & ~63u)));
#endif
	bool have = false; // this lane holds a ray (in flight, or finished and not yet emitted)
	bool more = true;  // items are left for this wave: in its own range or in the launch's queue (wave-uniform)
	bool global_more = true;
	uint32_t w_next = 0, w_end = 0; // the wave's own range of claimed items (wave-uniform)
	// Claims from the launch's counter, their size and what a launch's end looks like: see the refill below.  History of the
	// sizing (MI355X): fixed 256-item claims per wave, C3 2.84 ms; fixed 64, 2.57 ms but the room's 2 M primary rays 0.52 instead
	// of 0.31 ms (32 K atomics); round 2's rule (halve after 128 scheduling rounds on one claim, double within 40, between 64 and
	// 256): C3 2.69-2.75 ms, room 0.30 ms, 33 M short any-hit rays 1.63 ms of which 1.04 ms were the 130 K atomics; round 3 (sizes
	// from the measured rate, shared range per workgroup): C3 2.52 ms, those any-hit rays 1.30 ms, the test1 frame 0.235 instead
	// of 0.285 ms (profiles/ab_r03.md).
	const uint32_t resident_waves = gridDim.x * (kBlock / 64u);
	uint32_t seen = 0; // the counter as of this wave's last claim
	uint64_t t_first = 0;     // when this wave first claimed (wall_clock64, 100 MHz) ...
	uint32_t sum_claimed = 0; // ... and the items it has claimed since (without that first claim, which went out at once)
	uint32_t guard = 0;
#ifdef RTMI_TAIL_DIAG // diagnostic builds (tools/tail_diag.py): when each wave started, found its queue dry, and ended (100 MHz ticks)
	const unsigned long long diag_t0 = wall_clock64();
	unsigned long long diag_dry = 0;
	uint32_t diag_rounds_after = 0, diag_busy_after = 0;
#endif
	// End of the launch (closest hit only): a wave whose queue has run dry spreads the rays it still holds over its idle lanes.
	// A busy lane gives away the BOTTOM entry of its stack (the subtree it would reach last: the far side of a split near the
	// root) to an idle lane, which walks it with a copy of the ray and of the best hit so far; the entry is replaced by
	// kSentinel, which the giver pops last of all and which ends its walk like an empty stack does.  A helper that finishes is
	// folded into the lane that owns the item (smaller t wins, equal t -> smaller original id: the rule of the triangle step, so
	// the record is the sequential one whatever the split) and is free again; the owner writes the record when it and all its
	// helpers are done.  Helpers do not see hits found after they left, so they may visit subtrees the sequential walk would
	// have culled — on lanes that had nothing to do.
	constexpr bool SPLIT = kSplit && !ANY;
	int sb = 0;            // stack entries [0, sb) of this lane were given away
	uint32_t owner = lane; // the lane that holds the item this lane works for
	int nhelp = 0;         // helpers at work for this lane's item
	auto start = [&]() { // R holds a new ray: reset the lane's traversal state
		sb = 0; owner = lane; nhelp = 0;
		h.u = h.v = 0.f; h.pos = RTMI_MISS;
		best = R.tmax;
		tbest = f32_above(best);
		r = make_rayw(S, UNIFORM_O ? uox : R.ox, UNIFORM_O ? uoy : R.oy, UNIFORM_O ? uoz : R.oz, R.dx, R.dy, R.dz);
		sp = 0;
		cur = S.root_ref;
	};
	while (true) {
		// Retire + refill together, and only when enough lanes wait for it (or nothing else is left to do):
		// emit/fetch can be long (form factors: sample points, cosines, a division), so they must not run
		// for one lane at a time.
		if (SPLIT && !more) {
			// helpers that are done: fold their hit into the owner's, free the lane
			unsigned long long hf = __ballot(have && owner != lane && cur == kSentinel);
			while (hf != 0ull) {
				const int k = __ffsll((long long)hf) - 1;
				hf &= hf - 1ull;
				const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)owner, k);
				const uint32_t pk = (uint32_t)__builtin_amdgcn_readlane((int)h.pos, k);
				const double tk_ = readlane_f64(best, k);
				const float uk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(h.u), k));
				const float vk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(h.v), k));
				if (lane == o) {
					--nhelp;
					if (pk != RTMI_MISS && pk != h.pos &&
					    (tk_ < best || (tk_ == best && h.pos != RTMI_MISS && S.tris[pk].orig < S.tris[h.pos].orig))) {
						best = tk_;
						tbest = f32_above(best);
						h.u = uk; h.v = vk; h.pos = pk;
					}
				}
				if (lane == (uint32_t)k) { have = false; owner = lane; }
			}
			// helpers at work take over the owner's best hit when it is nearer than theirs (the owner walks the near side: what it
			// finds culls most of what the helpers were given)
			if (__ballot(have && owner != lane) != 0ull) {
				const double ob = shfl_f64(best, owner);
				const uint32_t op = (uint32_t)__shfl((int)h.pos, (int)owner, 64);
				if (have && owner != lane && ob < best) {
					best = ob;
					tbest = f32_above(best);
					h.pos = op; // u, v stay stale: a record copied from the owner never travels back (the fold skips equal positions)
				}
			}
			// idle lanes take subtrees from busy ones: the k-th idle lane from the k-th lane that has something to give
			const unsigned long long idle = __ballot(!have);
			const bool can_give = have && cur != kSentinel && sp > sb;
			const unsigned long long don = __ballot(can_give);
			if (idle != 0ull && don != 0ull) {
				const uint32_t n_pairs = min((uint32_t)__popcll(idle), (uint32_t)__popcll(don));
				const uint32_t rank_i = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
				const uint32_t rank_d = __builtin_amdgcn_mbcnt_hi((uint32_t)(don >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)don, 0u));
				int e = 0;
				if (can_give && rank_d < n_pairs) {
					e = stack[sb * kBlock];
					stack[sb * kBlock] = kSentinel;
					++sb;
				}
				const bool take = !have && rank_i < n_pairs;
				const int src = take ? (int)nth_set_bit64(don, rank_i) : (int)lane;
				const int pe = __shfl(e, src, 64);
				const uint32_t po = (uint32_t)__shfl((int)owner, src, 64);
				const uint32_t pp = (uint32_t)__shfl((int)h.pos, src, 64);
				const float pu = __shfl(h.u, src, 64), pv = __shfl(h.v, src, 64);
				const double pb = shfl_f64(best, src);
				const double rdx = shfl_f64(R.dx, src), rdy = shfl_f64(R.dy, src), rdz = shfl_f64(R.dz, src), rtm = shfl_f64(R.tmax, src);
				double rox = 0.0, roy = 0.0, roz = 0.0;
				if (!UNIFORM_O) { rox = shfl_f64(R.ox, src); roy = shfl_f64(R.oy, src); roz = shfl_f64(R.oz, src); }
				if (take) {
					R.dx = rdx; R.dy = rdy; R.dz = rdz; R.tmax = rtm;
					if (!UNIFORM_O) { R.ox = rox; R.oy = roy; R.oz = roz; }
					start();
					have = true;
					owner = po;
					best = pb;
					tbest = f32_above(best);
					h.u = pu; h.v = pv; h.pos = pp;
					cur = pe;
				}
				unsigned long long nh = __ballot(take); // the owners count their new helpers
				while (nh != 0ull) {
					const int k = __ffsll((long long)nh) - 1;
					nh &= nh - 1ull;
					if (lane == (uint32_t)__builtin_amdgcn_readlane((int)owner, k)) ++nhelp;
				}
			}
		}
		const bool fin = have && cur == kSentinel && (!SPLIT || (owner == lane && nhelp == 0));
		const int n_fin = __popcll(__ballot(fin)), n_free = more ? __popcll(__ballot(!have)) : 0;
		const int n_busy = __popcll(__ballot(have && cur != kSentinel));
		if (n_fin + n_free >= thresh || (n_busy == 0 && n_fin + n_free > 0) || (SPLIT && !more && n_fin > 0)) {
			if (fin) {
				h.t = h.pos != RTMI_MISS ? best : 0.0;
				if (emit(item, h, R)) start(); // the item wants another ray (R refilled): same lane carries on
				else have = false;
			}
			if (more) {
				const unsigned long long idle = __ballot(!have);
				const uint32_t nidle = (uint32_t)__popcll(idle);
				if (nidle > 0) {
					guard = 0; // the budget is per ray: a wave that keeps starting new rays keeps going
					// Items come from the wave's own range [w_next, w_end); when that runs short the wave takes more from the
					// range its workgroup shares, and when that is empty a whole claim from the launch's counter (one atomic
					// per claim, not per refill: thousands of waves adding to one address was measured to serialise).
					const uint32_t left = w_end - w_next;
					uint32_t fresh = 0, fresh_end = 0;
					bool got = false;
					bool pool_left = false;
					if (kPool) {
						// What a wave claims from the launch's counter goes into a range the workgroup's waves share (one 64-bit word in
						// LDS: next | end << 32, taken from by compare-and-swap).  A wave takes kPoolTake = 64 consecutive items from it at
						// a time into its own range [w_next, w_end) and refills its lanes from there (one 8 x 8 square of primary rays,
						// the rays of neighbouring hit points: a wave's lanes stay on neighbouring rays; refills straight from the shared
						// range, 16 items here and 16 there, were measured 8 % slower on the fused soft-shadow and AO queues).  When the
						// queue runs dry a workgroup is left with at most one claim for its 256 lanes and a wave with less than one ray
						// per lane — not with one claim per wave.
						if (left < nidle) {
							const uint32_t leader = (uint32_t)(__ffsll((long long)idle) - 1);
							const uint32_t need = nidle - left;
							uint32_t g_base = 0, g_n = 0, flags = 0, seen_now = seen; // flags: 1 = queue dry, 2 = pool not empty afterwards
							// Claim size (pool form).  A claim is an atomic on ONE address: about 8 ns each, one after the other, for the
							// whole device.  A workgroup knows the rate at which it uses items (what it claimed since its first claim over
							// the time since then, 100 MHz clock) and sizes its claims so that all resident workgroups claiming at its pace
							// keep the counter busy a quarter of the time at most: the power of two nearest to rate x gridDim.x x 32 ns,
							// between 64 and max_chunk.  (Doubling / halving on the last interval alone wandered up to the cap: measured.)
							// That lands cheap rays (short any-hit rays: 33 M in a millisecond) on the cap (512: larger claims gained there and
							// lost on the frames' smaller launches, profiles/ab_r03.md) and the soup's primary rays on 128.  Towards the end of the queue the size follows what is left: half an even share per workgroup,
							// but not below a quarter of the paced size (the last claims may crowd the counter, briefly).
							if (lane == leader) {
								for (;;) {
									const unsigned long long v = __hip_atomic_load(pool, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
									const uint32_t nx = (uint32_t)v, en = (uint32_t)(v >> 32);
									if (nx < en) {
										const uint32_t m = min(max(need, kPoolTake), en - nx);
										if (atomicCAS(pool, v, (unsigned long long)(nx + m) | ((unsigned long long)en << 32)) != v) continue;
										g_base = nx; g_n = m;
										if (nx + m < en) flags |= 2u;
										break;
									}
									if (!global_more) break;
									if (*pool_seen >= n_items) { flags |= 1u; break; } // another wave of the workgroup found the queue dry
									const uint64_t now = wall_clock64();
									const uint64_t t0 = *pool_t;   // the workgroup's first claim (its items went out at once: they fill the lanes)
									const uint32_t sum = *pool_sum; // items claimed after that one and used up by now
									uint32_t cc = tune.chunk;
									if (t0 != 0ull && sum != 0u) {
										const uint64_t dt = now > t0 ? now - t0 : 1ull;
										const uint64_t want = (uint64_t)sum * gridDim.x * 16ull / (dt * 5ull); // items used per tick x (gridDim.x x 3.2 ticks)
										const uint32_t w = (uint32_t)min((unsigned long long)tune.max_chunk, max(64ull, (unsigned long long)(want + (want >> 1))));
										cc = 1u << (31 - __clz((int)w)); // the power of two nearest to `want`
									}
									const uint32_t seen_wg = *pool_seen;
									const uint32_t rem = seen_wg < n_items ? n_items - seen_wg : 0u;
									const uint32_t chunk = min(cc, max(max(32u, cc >> kTailShift), rem / (2u * gridDim.x)));
									const uint32_t base = (uint32_t)atomicAdd(counter, (int)chunk);
									seen_now = base;
									*pool_seen = base; // the other waves of the workgroup learn how far the queue has got
									if (t0 == 0ull) *pool_t = now;
									else *pool_sum = sum + chunk;
									flags |= 4u; // claimed from the launch's counter
									if (base >= n_items) { flags |= 1u; break; }
									const uint32_t cend = base + chunk < n_items ? base + chunk : n_items;
									const uint32_t m = min(max(need, kPoolTake), cend - base);
									g_base = base; g_n = m;
									if (base + m < cend) {
										if (atomicCAS(pool, v, (unsigned long long)(base + m) | ((unsigned long long)cend << 32)) == v) flags |= 2u;
										else g_n = cend - base; // somebody else published first: the whole claim stays with this wave
									}
									break;
								}
							}
							fresh = (uint32_t)__builtin_amdgcn_readlane((int)g_base, (int)leader);
							fresh_end = fresh + (uint32_t)__builtin_amdgcn_readlane((int)g_n, (int)leader);
							flags = (uint32_t)__builtin_amdgcn_readlane((int)flags, (int)leader);
							seen = (uint32_t)__builtin_amdgcn_readlane((int)seen_now, (int)leader);
							got = fresh_end > fresh;
							if (flags & 1u) global_more = false;
							pool_left = (flags & 2u) != 0u;
							if (!have) {
								const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
								uint32_t idx = 0;
								bool ok = false;
								if (below < left) { idx = w_next + below; ok = true; }
								else if (got) { idx = fresh + (below - left); ok = idx < fresh_end; }
								if (ok && fetch(idx, R)) {
									item = idx;
									have = true;
									start();
								}
							}
							w_next = fresh + need < fresh_end ? fresh + need : fresh_end; // the rest of the piece is this wave's own
							w_end = fresh_end;
						} else {
							if (!have) {
								const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
								if (fetch(w_next + below, R)) {
									item = w_next + below;
									have = true;
									start();
								}
							}
							w_next += nidle;
							pool_left = true; // not looked at: assume so
						}
						more = global_more || w_next < w_end || pool_left;
					} else {
					if (left < nidle && global_more) {
						const uint32_t leader = (uint32_t)(__ffsll((long long)idle) - 1);
						uint32_t base = 0;
						// Claim size.  A claim is an atomic on ONE address: about 8 ns each, one after the other, for the whole device
						// (33 M short any-hit rays in 256-item claims: 130 K atomics = 1.04 ms of a 1.6 ms launch).  A wave knows the rate
						// at which it uses items (what it claimed since its first claim over the time since then, 100 MHz clock) and sizes
						// its claims so that all resident waves claiming at its pace keep the counter busy a quarter of the time at most:
						// the power of two nearest to rate x resident_waves x 32 ns, between 64 (one 8 x 8 square) and max_chunk.  Towards
						// the end of the queue the size follows what is left — half an even share per wave — but not below a quarter of
						// the paced size (the last claims may crowd the counter, briefly).
						const uint64_t now = wall_clock64();
						uint32_t cc = tune.chunk;
						if (t_first != 0ull && sum_claimed != 0u) {
							const uint64_t dt = now > t_first ? now - t_first : 1ull;
							const uint64_t want = (uint64_t)sum_claimed * resident_waves * 16ull / (dt * 5ull); // items per tick x (resident_waves x 3.2 ticks)
							const uint32_t w = (uint32_t)min((unsigned long long)tune.max_chunk, max(64ull, (unsigned long long)(want + (want >> 1))));
							cc = 1u << (31 - __clz((int)w));
						}
						const uint32_t rem = seen < n_items ? n_items - seen : 0u;
						const uint32_t chunk = min(cc, max(max(tune.tail_chunk, cc >> 2), rem / (2u * resident_waves)));
						if (t_first == 0ull) t_first = now;
						else sum_claimed += chunk;
						if (lane == leader) base = (uint32_t)atomicAdd(counter, (int)chunk);
						fresh = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)leader);
						seen = fresh;
						if (fresh >= n_items) {
							global_more = false;
						} else {
							got = true;
							fresh_end = fresh + chunk < n_items ? fresh + chunk : n_items;
						}
					}
					if (!have) {
						const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
						uint32_t idx = 0;
						bool ok = false;
						if (below < left) { idx = w_next + below; ok = true; }
						else if (got) { idx = fresh + (below - left); ok = idx < fresh_end; }
						if (ok && fetch(idx, R)) {
							item = idx;
							have = true;
							start();
						}
					}
					if (left >= nidle) {
						w_next += nidle;
					} else if (got) {
						w_next = fresh + (nidle - left) < fresh_end ? fresh + (nidle - left) : fresh_end;
						w_end = fresh_end;
					} else {
						w_next = w_end;
					}
					more = global_more || w_next < w_end;
					}
				}
			}
		}
		if (cur < 0 && cur != kSentinel && cur != kLeaf) { // a leaf reference came up (child, pop or a single-leaf root): unpack it
			const uint32_t code = ~(uint32_t)cur;
			tk = code >> 3;
			tend = tk + (code & 7u) + 1u;
			cur = kLeaf;
		}
		if (++guard > S.guard) { give_up(S); break; } // a wave that cannot finish must still end: never reached on valid input
#ifdef RTMI_TAIL_DIAG
		if (!more) {
			if (diag_dry == 0) diag_dry = wall_clock64();
			++diag_rounds_after;
			diag_busy_after += (uint32_t)__popcll(__ballot(have && cur != kSentinel));
		}
#endif
		// Vote: the wave runs ONE node step or ONE triangle step per iteration, whichever has more lanes
		// waiting (a triangle step costs about twice a node step).  Lanes of the other kind simply wait, so
		// no lane ever runs ahead speculatively and the visited set is exactly the sequential one.
		const int n_node = __popcll(__ballot(cur >= 0)), n_leaf = __popcll(__ballot(cur == kLeaf));
		if (n_node + n_leaf == 0) { // nobody can step: finished lanes retire at the top of the next round
			if (!more && __ballot(have) == 0ull) {
				if (kPool) { // a wave of this workgroup may have shared the rest of an earlier claim since this one last looked
					const unsigned long long v = __hip_atomic_load(pool, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
					if ((uint32_t)v < (uint32_t)(v >> 32)) { more = true; continue; }
				}
				break;
			}
			continue;
		}
		if (n_node * vote_node >= n_leaf * vote_leaf) {
			// kNodeSteps node steps per round: the retire/refill test and the vote are paid once for them
#pragma unroll
			for (int rep = 0; rep < kNodeSteps; ++rep) {
				if (COUNT && __ballot(cur >= 0) != 0ull) { // who sits out this node step, and why
					if (cur < 0 && cur != kSentinel) ++w->wait_leaf;     // holds a leaf, waits for a triangle round
					else if (cur == kSentinel) ++w->wait_idle;            // finished or without a ray, waits for the refill
				}
#if RTMI_PAIR_FETCH
				if (__ballot(cur >= 0) == 0ull) break; // nobody has a node left this round (wave-uniform)
				uint4 na, nb, nc;
				pair_fetch(S, cur, lane, na, nb, nc);
				if (cur >= 0) {
					if (COUNT) { ++w->nodes; w->node_slots += wave_slots(); }
					cur = node_step_abc(S, r, tbest, na, nb, nc, sp, stack);
				}
#else
				if (cur >= 0) {
					if (COUNT) { ++w->nodes; w->node_slots += wave_slots(); }
#if RTMI_DRAIN_PREFETCH
					cur = node_step_prefetching(S, r, tbest, cur, sp, stack, !more, pf_land);
#else
					cur = node_step(S, r, tbest, cur, sp, stack, top, n_top);
#endif
				}
#endif
			}
		} else {
			if (cur == kLeaf) {
				if (COUNT) { ++w->tests; w->test_slots += wave_slots(); }
				const GTri T = load_tri(S.tris + tk); // the whole 80-B record in one round trip
				double u = 0.0, v = 0.0;
				const double t = mt_exact(UNIFORM_O ? uox : R.ox, UNIFORM_O ? uoy : R.oy, UNIFORM_O ? uoz : R.oz, R.dx, R.dy, R.dz, T, u, v);
				bool done = false;
				if (ANY) {
					if (t > 0.0 && t < best) { // best == tmax throughout
						best = t;
						h.u = (float)u; h.v = (float)v; h.pos = tk;
						done = true;
					}
				} else {
					// strict minimum; equal t -> smaller original id (the rare tie reads the holder's id back)
					if (t > 0.0 && (best > t || (t == best && h.pos != RTMI_MISS && T.orig < S.tris[h.pos].orig))) {
						best = t;
						tbest = f32_above(best);
						h.u = (float)u; h.v = (float)v; h.pos = tk;
					}
				}
				++tk;
				if (done || (tk == tend && sp == 0)) {
					cur = kSentinel;
				} else if (tk == tend) {
					--sp;
					cur = stack[sp * kBlock];
				}
			}
		}
	}
#ifdef RTMI_TAIL_DIAG
	if (!ANY && UNIFORM_O && lane == 0) {
		unsigned long long *o = g_tail_diag + 5u * ((blockIdx.x * (kBlock / 64u) + (threadIdx.x >> 6)) & 8191u);
		o[0] = diag_t0; o[1] = diag_dry; o[2] = wall_clock64(); o[3] = diag_rounds_after; o[4] = diag_busy_after;
	}
#endif
}

// ---------------------------------------------------------------- reference vector helpers (double)
struct D3 { double x, y, z; };
__device__ __forceinline__ D3 d3(double x, double y, double z) { D3 r = {x, y, z}; return r; }
__device__ __forceinline__ D3 operator+(D3 a, D3 b) { return d3(a.x + b.x, a.y + b.y, a.z + b.z); }     // vec3.h:22
__device__ __forceinline__ D3 operator-(D3 a, D3 b) { return d3(a.x - b.x, a.y - b.y, a.z - b.z); }     // vec3.h:26
__device__ __forceinline__ D3 operator*(double s, D3 a) { return d3(s * a.x, s * a.y, s * a.z); }       // vec3.h:34
__device__ __forceinline__ double dot(D3 a, D3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }   // vec3.h:42
// Three IEEE divisions by ONE denominator (vec3.h:46), bit for bit what `a.x / c, a.y / c, a.z / c` give.  The compiler expands a
// double division into  d' = div_scale(c) ; r = rcp(d') refined by four FMAs ; n' = div_scale(n) ; q = n'·r ; e = n' − d'·q ;
// div_fmas(e, r, q) ; div_fixup  — twelve instructions, one of them the 17-cycle v_rcp_f64 — and repeats all of it per
// numerator, because d' may depend on the numerator (the scaling only acts at the ends of the exponent range).  Here the
// reciprocal and its refinement are computed once when the three scaled denominators are the same bits — always, outside
// those ends — and the plain divisions run otherwise: the same instructions on the same operands, so the same quotients
// (tests/test_gpu_parity.py: test_shared_reciprocal_division_is_the_plain_division).  Saves 2 v_rcp_f64 + 10 FMAs per call.
#ifndef RTMI_PLAIN_DIV3
__device__ __forceinline__ D3 div3(D3 a, double c) {
	bool s0, s1, s2, unused;
	const double dx = __builtin_amdgcn_div_scale(a.x, c, false, &unused);
	const double dy = __builtin_amdgcn_div_scale(a.y, c, false, &unused);
	const double dz = __builtin_amdgcn_div_scale(a.z, c, false, &unused);
	if (__double_as_longlong(dx) == __double_as_longlong(dy) && __double_as_longlong(dx) == __double_as_longlong(dz)) {
		double r = __builtin_amdgcn_rcp(dx);
		const double f0 = __builtin_fma(-dx, r, 1.0);
		r = __builtin_fma(r, f0, r);
		const double f2 = __builtin_fma(-dx, r, 1.0);
		r = __builtin_fma(r, f2, r);
		const double nx = __builtin_amdgcn_div_scale(a.x, c, true, &s0);
		const double ny = __builtin_amdgcn_div_scale(a.y, c, true, &s1);
		const double nz = __builtin_amdgcn_div_scale(a.z, c, true, &s2);
		const double qx = nx * r, qy = ny * r, qz = nz * r;
		const double ex = __builtin_fma(-dx, qx, nx), ey = __builtin_fma(-dx, qy, ny), ez = __builtin_fma(-dx, qz, nz);
		return d3(__builtin_amdgcn_div_fixup(__builtin_amdgcn_div_fmas(ex, r, qx, s0), c, a.x),
		          __builtin_amdgcn_div_fixup(__builtin_amdgcn_div_fmas(ey, r, qy, s1), c, a.y),
		          __builtin_amdgcn_div_fixup(__builtin_amdgcn_div_fmas(ez, r, qz, s2), c, a.z));
	}
	return d3(a.x / c, a.y / c, a.z / c);
}
#else
__device__ __forceinline__ D3 div3(D3 a, double c) { return d3(a.x / c, a.y / c, a.z / c); }
#endif
__device__ __forceinline__ D3 operator/(D3 a, double c) { return div3(a, c); }                            // vec3.h:46
__device__ __forceinline__ double len2(D3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }              // vec3.h:14
// a / sqrt(d2) with d2 = len2(a), bit for bit what the compiler's square root and div3 give — for the form-factor kernel's sample
// loop, where the two are a fifth of the common path.  Both expansions carry range handling that acts only at the ends of the
// exponent range: the root rescales operands below 2^-767 and passes 0 and inf through; a division's v_div_scale / v_div_fmas /
// v_div_fixup rescale when an operand or the quotient comes near a denormal or the two exponents lie 768 apart, and patch up zeros,
// infinities and NaNs.  When EVERY lane of the wave has 2^-600 <= d2 <= 2^600 and all three |components| >= 2^-300 (then the
// length lies in 2^±300, the quotients in [2^-601, 1]: nothing is rescaled, nothing is special), the wave runs the bare sequences
// — the same instructions on the same operands minus the ones that would do nothing: 24 instead of 43.  Otherwise the plain forms.
// (tests/test_gpu_parity.py: test_unit_vector_fast_path_is_the_plain_root_and_division)
__device__ __forceinline__ D3 unit_of_len2(D3 a, double d2) {
#ifndef RTMI_PLAIN_UNIT
	double m;
	asm("v_min_f64 %0, |%1|, |%2|" : "=v"(m) : "v"(a.x), "v"(a.y));
	asm("v_min_f64 %0, %1, |%2|" : "=v"(m) : "v"(m), "v"(a.z));
	if ((__ballot(d2 >= 0x1.0p-600) & __ballot(d2 <= 0x1.0p600) & __ballot(m >= 0x1.0p-300)) == __ballot(true)) {
		// the root (llvm's expansion without the scaling by 2^256 / 2^-128 and the class test)
		const double y = __builtin_amdgcn_rsq(d2);
		double g = d2 * y, h = y * 0.5;
		const double r0 = __builtin_fma(-h, g, 0.5);
		g = __builtin_fma(g, r0, g);
		h = __builtin_fma(h, r0, h);
		const double e0 = __builtin_fma(-g, g, d2);
		g = __builtin_fma(e0, h, g);
		const double e1 = __builtin_fma(-g, g, d2);
		g = __builtin_fma(e1, h, g);
		// three divisions by it (div3's shared form without the scalings and the fix-up)
		double r = __builtin_amdgcn_rcp(g);
		const double f0 = __builtin_fma(-g, r, 1.0);
		r = __builtin_fma(r, f0, r);
		const double f1 = __builtin_fma(-g, r, 1.0);
		r = __builtin_fma(r, f1, r);
		const double qx = a.x * r, qy = a.y * r, qz = a.z * r;
		return d3(__builtin_fma(__builtin_fma(-g, qx, a.x), r, qx), __builtin_fma(__builtin_fma(-g, qy, a.y), r, qy), __builtin_fma(__builtin_fma(-g, qz, a.z), r, qz));
	}
#endif
	return a / sqrt(d2);
}
__device__ __forceinline__ D3 unit(D3 a) {                                                              // vec3.h:98
	const double d = len2(a);
#ifdef RTMI_UNIT_FAST // experiment (profiles/ab_r03.md): the bare root and divisions for every Unit() of the kernels
	return d > 0.0 ? unit_of_len2(a, d) : d3(0, 0, 0);
#else
	return d > 0.0 ? a / sqrt(d) : d3(0, 0, 0);
#endif
}


// ---------------------------------------------------------------- wavefront helpers shared by the ray generators
constexpr double kPiRef = 3.14159265358979; // Render2/base.h:52

__device__ __forceinline__ D3 cross(D3 a, D3 b) { // vec3.h:50
	return d3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ uint64_t wf_mix64(uint64_t z) {
	z += 0x9E3779B97F4A7C15ull;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}

// hit.point of updateHitInfo (object.cpp:23): origin + (direction·((triVert1 − origin)·n)) / (n·direction)
__device__ __forceinline__ D3 hit_point(const rtmi_ray &r, const double *v1, D3 n) {
	const D3 o = d3(r.o[0], r.o[1], r.o[2]), d = d3(r.d[0], r.d[1], r.d[2]);
	return o + (dot(d3(v1[0], v1[1], v1[2]) - o, n) * d) / dot(n, d);
}

__device__ __forceinline__ void inactive(rtmi_ray &q) {
	q.o[0] = q.o[1] = q.o[2] = 0.0;
	q.d[0] = q.d[1] = 0.0; q.d[2] = 1.0;
	q.tmax = RTMI_INFINITY;
	q.reserved = 1.0;
}
__device__ __forceinline__ void set_ray(rtmi_ray &q, D3 org, D3 dir, double tmax) {
	q.o[0] = org.x; q.o[1] = org.y; q.o[2] = org.z;
	q.d[0] = dir.x; q.d[1] = dir.y; q.d[2] = dir.z;
	q.tmax = tmax;
	q.reserved = 0.0;
}

// findOcclusionNew (shader.cpp:140-232): the hemisphere frame of hit.normal (:153-171) ...
__device__ __forceinline__ void ao_frame(D3 nrm, D3 &X, D3 &Y, D3 &Z) {
	Z = unit(nrm); // :153
	D3 t = Z;
	const double cz = fabs(Z.z), cx = fabs(Z.x), cy = fabs(Z.y);
	if (cz <= cx) { // :160-168, the reference's own branch structure
		if (cz <= cy) t.z = 1;
	} else if (cx <= cz) {
		if (cx <= cy) t.x = 1;
	} else t.y = 1;
	X = unit(cross(t, Z)); // :170
	Y = cross(Z, X);       // :171
}
// ... and one ray in it: the square point (a, b) comes from the 64-bit hash hsh (the reference draws it from an
// unseeded mt19937, :148-150,177).
__device__ __forceinline__ void ao_ray_in_frame(D3 P, D3 X, D3 Y, D3 Z, double R, uint64_t hsh, D3 &org, D3 &dir) {
	const double a = (2.0 * ((double)(hsh >> 11) * 0x1.0p-53)) - 1.0; // dis(gen) − 1, dis = uniform(0, 2)
	const double b = (2.0 * ((double)(wf_mix64(hsh) >> 11) * 0x1.0p-53)) - 1.0;
	double phi, r;
	if (a > -b) { // :182-204
		if (a > b) { r = a; phi = (kPiRef / 4.0) * (b / a); }
		else { r = b; phi = (kPiRef / 4.0) * (2.0 - (a / b)); }
	} else {
		if (a < b) { r = -a; phi = (kPiRef / 4.0) * (4.0 + (b / a)); }
		else { r = -b; phi = (b != 0) ? (kPiRef / 4.0) * (6.0 - (a / b)) : 0; }
	}
	const double u = r * cos(phi), v = r * sin(phi);               // :206-207
	const D3 base = (u * X) + (v * Y);                             // :210-211
	const double lb = sqrt(len2(base));
	const double leg = (R * R) - (lb * lb);                        // :212
	dir = unit(((leg / R) * Z) + base);                            // :214,218
	org = P + (.0001 * dir);                                       // :220
}
__device__ __forceinline__ void ao_ray(D3 P, D3 nrm, double R, uint64_t hsh, D3 &org, D3 &dir) {
	D3 X, Y, Z;
	ao_frame(nrm, X, Y, Z);
	ao_ray_in_frame(P, X, Y, Z, R, hsh, org, dir);
}

} // namespace rtmi
