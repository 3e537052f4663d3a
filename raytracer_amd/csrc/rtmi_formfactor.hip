// librtmi — radiosity: the form-factor build (include/rtmi.h: rtmi_form_factors_dev), gfx950 only.
//
// Reference: Radiosity::findFormFactor's fill loops (Render2/radiosity.cpp:379-403) over findFormFactorTerm (:305-333),
// isVisibleXiToXj (:231-262) and findFormFactorTermWithLight (:264-303).  Every quantity that decides a bit of F is
// evaluated in double in the reference's operation order (the sample points come from the counter-based generator of
// rtmi_rad_device.hpp, which the oracle restates), so F equals the oracle's bit for bit.
//
// What is new here is HOW the closest hit of a visibility ray is found.  A form-factor ray is not an arbitrary ray: it
// leaves patch i and is aimed at a point of patch j, and both are leaves of the hierarchy whose root-to-leaf paths are
// known.  Every other triangle lies in exactly one subtree hanging off one of those two paths (a "sibling" of the path).
// So instead of descending from the root — about forty box-pair tests per ray in a closed room, nearly all of them spent
// rediscovering the two leaves — the kernel
//   1. tests triangle j first (exact Möller–Trumbore): that bounds the ray at its target,
//   2. tests the ray against the sibling boxes along the two paths, and walks into a sibling's subtree only where the box
//      is hit (the ordinary stack traversal, rtmi_device.hpp: node_step),
//   3. reads the verdict off the closest hit exactly as the reference does.
// Skipping a box test is always safe (nothing is culled by it), every sibling box is the same conservative box the
// root traversal would have tested, and the closest hit with its tie rule (smaller triangle id) does not depend on the
// order in which triangles are met: the hit record — hence F — is the root traversal's, bit for bit.  When triangle j
// itself is not hit (a sample point on its very edge) the ray falls back to the root traversal.
// The ten sample rays of a pair (i, j) are nearly parallel, so step 2 is done once per pair against the BEAM between the
// two patches' bounds (one conservative slab test per sibling box); the per-ray exact test runs only for the survivors —
// in a room, the other triangle of each quad.
//
// Work layout: a wave owns 64 patches j that are neighbours in depth-first leaf order (coherent targets, and their paths
// share their upper parts) and walks a band of rows i; everything that depends on i alone — vertices, normal, path — is
// wave-uniform and lives in scalar registers.  A launch is (groups of j) × (bands of i) workgroups; the tables one
// workgroup touches (its four groups' paths, the band's paths) are a few hundred KB and stay in its XCD's L2.
#include "rtmi_rad_device.hpp"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unordered_map>
#include <vector>

namespace rtmi {

// One sibling along a root-to-leaf path: its fp32 bounds (the quantised box of the binary node, which contains the
// triangles' bounds) and its reference (inner node >= 0, leaf < 0 as in QNode) in both hierarchies.  32 B = two 16-byte loads.
struct __attribute__((aligned(16))) ChainEntry {
	float lo[3];
	int32_t ref;  // in the binary hierarchy (the beam walk)
	float hi[3];
	int32_t wide; // = ref; with -DRTMI_FF_WIDE_WALK=1 (experiment, not kept) where the same subtree sits in the 4-wide hierarchy (Wide4::cover)
};
// Per patch (original id): where its path's siblings start in the chain array, the path itself (bit L = went right at
// level L), its length (< 32) with the position of its triangle in the leaf-ordered array (< kMaxFirst = 2^27) in one word,
// and which of ITS siblings lie wholly behind (or in) ITS plane (bit L = the sibling at level L; build_ff_tables).  The flag
// belongs to the patch, not to the sibling: two patches of one 64-patch group share their upper siblings but not their planes
// (a group that straddles an edge of a box, a stair, a curved surface), so each lane reads its own mask.
struct __attribute__((aligned(16))) PatchMeta {
	uint32_t chain_off, path, depth_leafpos, behind;
};
__host__ __device__ __forceinline__ uint32_t meta_depth(const PatchMeta &m) { return m.depth_leafpos >> 27; }
__host__ __device__ __forceinline__ uint32_t meta_leafpos(const PatchMeta &m) { return m.depth_leafpos & 0x7ffffffu; }
__host__ __device__ __forceinline__ uint32_t bits_below(uint32_t n) { return n >= 32u ? ~0u : ((1u << n) - 1u); } // levels 0 .. n-1
__host__ __device__ __forceinline__ uint32_t bits_above(uint32_t n) { return n >= 31u ? 0u : ~((2u << n) - 1u); }  // levels n+1 .. 31
struct FFTables {
	const ChainEntry *chain;
	const PatchMeta *meta;
	const uint32_t *dfs;
	uint32_t n_groups;
};

// Slab test against an fp32 box widened per axis by a pad folded into the addends: t(lo plane) = lo·i − (o + pad)·i,
// t(hi plane) = hi·i − (o − pad)·i (the pad dominates every rounding on the way).
struct RayS {
	float ix, iy, iz;
	float lx, ly, lz; // −(o + pad)·i
	float hx, hy, hz; // −(o − pad)·i
};
__device__ __forceinline__ RayS make_rays(float ox, float oy, float oz, float dx, float dy, float dz, float px, float py, float pz) {
	RayS r;
	r.ix = safe_rcp(dx); r.iy = safe_rcp(dy); r.iz = safe_rcp(dz);
	r.lx = -(ox + px) * r.ix; r.ly = -(oy + py) * r.iy; r.lz = -(oz + pz) * r.iz;
	r.hx = -(ox - px) * r.ix; r.hy = -(oy - py) * r.iy; r.hz = -(oz - pz) * r.iz;
	return r;
}
__device__ __forceinline__ bool slab_box(const float4 a, const float4 b, const RayS &r, float tmax) {
	const float tlx = fmaf(a.x, r.ix, r.lx), thx = fmaf(b.x, r.ix, r.hx);
	const float tly = fmaf(a.y, r.iy, r.ly), thy = fmaf(b.y, r.iy, r.hy);
	const float tlz = fmaf(a.z, r.iz, r.lz), thz = fmaf(b.z, r.iz, r.hz);
	const float tn = fmaxf(fmaxf(fminf(tlx, thx), fminf(tly, thy)), fmaxf(fminf(tlz, thz), 0.0f));
	const float tf = fminf(fminf(fmaxf(tlx, thx), fmaxf(tly, thy)), fminf(fmaxf(tlz, thz), tmax));
	return tn <= tf;
}

// Work counters of the form-factor kernel (diagnostic builds only: make EXTRA=-DRTMI_FF_COUNT; tools/ff_counts.py)
#ifdef RTMI_FF_COUNT
__device__ unsigned long long g_ff_counts[96];
// (n is evaluated by every active lane BEFORE the election: a ballot inside the elected lane's branch would see that lane alone)
#define FF_COUNT(k, n) do { const unsigned long long n_ = (unsigned long long)(n); if ((threadIdx.x & 63u) == (uint32_t)(__ffsll((long long)__ballot(1)) - 1)) atomicAdd(&g_ff_counts[k], n_); } while (0)
#define FF_COUNT_LANES(k, pred) FF_COUNT(k, __popcll(__ballot(pred)))
#else
#define FF_COUNT(k, n) do { } while (0)
#define FF_COUNT_LANES(k, pred) do { } while (0)
#endif

// Pair level: a sibling that is an inner node and whose box the beam meets is walked with the BEAM (at most kBeamBudget
// node visits, on the lane's own stack): if the beam meets no leaf's box below it, no ray of the pair can hit anything
// there, and the sibling is dropped for all samples of the pair.  This is what disposes of the big boxes at the top of a
// room's hierarchy (pieces of several flat walls in one box = a hollow volume the rays cross).
#ifndef RTMI_FF_BEAM_BUDGET
#define RTMI_FF_BEAM_BUDGET 3
#endif
constexpr int kBeamBudget = RTMI_FF_BEAM_BUDGET;
__device__ __forceinline__ bool beam_meets_leaf(const SceneDev &S, int ref, const RayS &beam, int *stack) {
	if (ref < 0) return true; // a leaf: its box was the test
	int sp = 0, cur = ref;
	for (int visits = 0; visits < kBeamBudget; ++visits) {
		FF_COUNT(80, 1);
		const QNode *np = S.nodes + cur;
		const uint4 a = np->a, b = np->b;
		auto box = [&](uint32_t wx, uint32_t wy, uint32_t wz) {
			const float4 lo = make_float4(fmaf((float)(wx & 0xffffu), S.qstep[0], S.qlo[0]), fmaf((float)(wy & 0xffffu), S.qstep[1], S.qlo[1]),
			                              fmaf((float)(wz & 0xffffu), S.qstep[2], S.qlo[2]), 0.f);
			const float4 hi = make_float4(fmaf((float)(wx >> 16), S.qstep[0], S.qlo[0]), fmaf((float)(wy >> 16), S.qstep[1], S.qlo[1]),
			                              fmaf((float)(wz >> 16), S.qstep[2], S.qlo[2]), 0.f);
			return slab_box(lo, hi, beam, 1.0f);
		};
		const bool h0 = box(a.x, a.y, a.z), h1 = box(a.w, b.x, b.y);
		const int r0 = (int)b.z, r1 = (int)b.w;
		if ((h0 && r0 < 0) || (h1 && r1 < 0)) return true;
		if (h0 && h1) { stack[sp * kBlock] = r1; ++sp; cur = r0; }
		else if (h0) cur = r0;
		else if (h1) cur = r1;
		else if (sp > 0) { --sp; cur = stack[sp * kBlock]; }
		else return false;
	}
	return true; // out of budget: keep the sibling
}

__device__ __forceinline__ uint32_t uniform_u32(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t x) {
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) x = max(x, (uint32_t)__shfl_xor((int)x, off, 64));
	return uniform_u32(x);
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t x) {
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) x = min(x, (uint32_t)__shfl_xor((int)x, off, 64));
	return uniform_u32(x);
}
__device__ __forceinline__ uint32_t wave_or_u32(uint32_t x) {
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) x |= (uint32_t)__shfl_xor((int)x, off, 64);
	return uniform_u32(x);
}

// fp32 bounds of a patch (three vertices, doubles): nearest rounding — the beam test's margin covers the half ulp
__device__ __forceinline__ void patch_bounds(const double *v, float lo[3], float hi[3]) {
#pragma unroll
	for (int a = 0; a < 3; ++a) {
		const float p = (float)v[a], q = (float)v[3 + a], r = (float)v[6 + a];
		lo[a] = fminf(p, fminf(q, r));
		hi[a] = fmaxf(p, fmaxf(q, r));
	}
}

__device__ __forceinline__ int step_from(const SceneDev &S, const RayF &r, float tbest, int cur, int &sp, int *stack) { return node_step(S, r, tbest, cur, sp, stack); }
__device__ __forceinline__ int step_from(const SceneDev &S, const RayW &r, float tbest, int cur, int &sp, int *stack) { return node_step_cover(S, r, tbest, cur, sp, stack); }

// The ordinary stack traversal, started from whatever a lane holds: `cur` (a node, a leaf reference, or
// kSentinel) and `sp` entries on its stack.  Closest hit with the tie rule; vote-scheduled like trace(): per iteration the
// wave takes ONE node step or ONE triangle step.
// RayW: through the 4-wide nodes, from references that may name some of a node's children (node_step_cover); RayF: the binary nodes.
template <class Ray>
__device__ __forceinline__ void trace_from(const SceneDev &S, const Ray &rf, double ox, double oy, double oz, double dx, double dy, double dz,
                                           double &best, uint32_t &pos, int cur, int sp, int *stack, uint32_t tag = 0u) {
	(void)tag; // diagnostic builds: which kind of call the steps are counted under
	float tbest = f32_above(best);
	uint32_t tk = 0, tend = 0;
	for (uint32_t guard = 0;; ++guard) {
		if (guard >= S.guard) { give_up(S); break; } // never reached on valid input
		if (cur < 0 && cur != kSentinel && cur != kLeaf) {
			const uint32_t code = ~(uint32_t)cur;
			tk = code >> 3;
			tend = tk + (code & 7u) + 1u;
			cur = kLeaf;
		}
		const int n_node = __popcll(__ballot(cur >= 0)), n_leaf = __popcll(__ballot(cur == kLeaf));
		if (n_node + n_leaf == 0) break;
		if (n_node >= n_leaf) {
			FF_COUNT(6, 1); FF_COUNT(11, n_node); FF_COUNT(74 + tag, 1);
			if (cur >= 0) cur = step_from(S, rf, tbest, cur, sp, stack);
		} else if (cur == kLeaf) {
			FF_COUNT(7, 1); FF_COUNT(12, n_leaf); FF_COUNT(76 + tag, 1);
			const GTri T = load_tri(S.tris + tk);
			double u = 0.0, v = 0.0;
			const double t = mt_exact(ox, oy, oz, dx, dy, dz, T, u, v);
			// strict minimum; equal t -> smaller original id (Render2/object.cpp:155 keeps the first of its in-order scan)
			if (t > 0.0 && (best > t || (t == best && pos != RTMI_MISS && T.orig < S.tris[pos].orig))) {
				best = t;
				tbest = f32_above(best);
				pos = tk;
			}
			++tk;
			if (tk == tend) {
				if (sp > 0) { --sp; cur = stack[sp * kBlock]; }
				else cur = kSentinel;
			}
		}
	}
}

// Render2/radiosity.cpp:264-303 (findFormFactorTermWithLight) for patch i
__device__ __forceinline__ double light_term(const SceneDev &S, const PatchesDev &P, uint32_t i, D3 C, uint32_t light_samples, uint64_t seed,
                                             uint32_t flags, int *stack) {
	const double *ei = P.verts + 9 * (size_t)i;
	const D3 ni = d3(P.normals[3 * (size_t)i], P.normals[3 * (size_t)i + 1], P.normals[3 * (size_t)i + 2]);
	const bool mean = (flags & RTMI_FF_LIGHT_MEAN) != 0;
	double ff = 0.0;
	// the reference zeroes its accumulator inside the loop (:276): only the last sample survives
	for (uint32_t k = mean ? 0u : light_samples - 1; k < light_samples; ++k) {
		const D3 x = sample_point(ei, seed, i, kLightStream, k, 0);
		const D3 dir = unit(C - x);
		const D3 org = x + (kEpsilon * ni);
		const double L = sqrt(len2(C - x));
		const double tmp = ((1.0 / (4.0 * kPi)) * (dot(dir, ni) / (L * L)));
		if (tmp > 0.0) { // only a positive term is ever added (:297-299): the ray is needed for those alone
			HitP h;
			trace<true, false, false>(S, org.x, org.y, org.z, dir.x, dir.y, dir.z, RTMI_INFINITY, stack, h);
			if (h.pos == RTMI_MISS) ff += tmp;
		}
	}
	return light_samples ? ff / ((double)light_samples) : 0.0;
}

// Column n_patches (the light) of rows [row0, row0 + n_rows), and the light's own row: one lane per row.
RTMI_TRAVERSAL_KERNEL void k_ff_light_column(SceneDev S, PatchesDev P, double cx, double cy, double cz, uint32_t light_samples, uint64_t seed,
                                             uint32_t flags, uint32_t row0, uint32_t n_rows, float *__restrict__ F, uint64_t ld) {
	extern __shared__ int stack_all[]; // rtmi_scene::stack_bytes_binary
	const uint32_t r = blockIdx.x * kBlock + threadIdx.x;
	if (r >= n_rows) return;
	const uint32_t i = row0 + r;
	double out = 0.0;
	if (i < P.n) out = light_term(S, P, i, d3(cx, cy, cz), light_samples, seed, flags, stack_all + threadIdx.x);
	F[(uint64_t)r * ld + P.n] = (float)out; // the light's own entry (:382) is 0
}

// 4 waves per SIMD: 124 of 128 registers, no scratch (what depends on j alone and is needed once per row — the store address,
// the area, the patch's bounds — is computed again per row through laundered indices instead of being held over the kernel;
// held, three of those values were spilled).  The kernel issues vector instructions in only 0.6 of the cycles at 3 waves
// — dependent double-precision chains — so the fourth wave pays: 34.2 -> 38.2 G pair-samples/s on the 65 712-patch box;
// 5 waves (96 registers, 35 spills) 31.4.
#ifndef RTMI_FF_WAVES
#define RTMI_FF_WAVES 4
#endif
#ifndef RTMI_FF_SKIP_P
#define RTMI_FF_SKIP_P 1
#endif
#ifndef RTMI_FF_SKIP_MT
#define RTMI_FF_SKIP_MT 1
#endif
#ifndef RTMI_FF_SAME_PLANE
#define RTMI_FF_SAME_PLANE 1
#endif
#ifndef RTMI_FF_REGROUP
#define RTMI_FF_REGROUP 1
#endif
#ifndef RTMI_FF_REGROUP_NUM // samples are dealt over the wave when that takes at most NUM/DEN as many passes as there are samples
#define RTMI_FF_REGROUP_NUM 1
#define RTMI_FF_REGROUP_DEN 2
#endif
#ifndef RTMI_FF_XCD_BLOCK
#define RTMI_FF_XCD_BLOCK 1
#endif
// Workgroup = ONE group of 64 patches j (blockIdx.x); its four waves take four consecutive bands of rows i, so the group's
// per-lane tables are fetched once per workgroup and shared through the L1.  Columns j < n_patches only; the light's row
// (i == n_patches) is zero (:391-393), the diagonal too (:382).
// stack: pend_cap + (the deeper hierarchy's path) + 1 entries per lane — sibling subtrees waiting, and the traversal inside one of them.
// (The tables come as separate __restrict__ pointers: only then does the compiler know that the stores to F cannot change
// them, and fetches everything that depends on i alone with scalar loads.)
RTMI_TRAVERSAL_KERNEL_W(RTMI_FF_WAVES) void k_form_factors(SceneDev S, const double *__restrict__ p_verts, const double *__restrict__ p_normals,
                                                           const double *__restrict__ p_area, uint32_t p_n, const ChainEntry *__restrict__ t_chain,
                                                           const PatchMeta *__restrict__ t_meta, const uint32_t *__restrict__ t_dfs, const uint4 *__restrict__ t_gconst,
                                                           uint32_t samples, uint64_t seed, uint32_t row0, uint32_t n_rows, uint32_t band_rows,
                                                           float margin, double graze_cos_i, double graze_cos_j, uint32_t planes_on, uint32_t pend_cap, uint32_t n_groups, float *__restrict__ F, uint64_t ld) {
	struct { const double *verts, *normals, *area; uint32_t n; } P = {p_verts, p_normals, p_area, p_n};
	struct { const ChainEntry *chain; const PatchMeta *meta; const uint32_t *dfs; } T = {t_chain, t_meta, t_dfs};
	extern __shared__ int stack_all[];
	int *stack = stack_all + threadIdx.x;
	const uint32_t lane = threadIdx.x & 63u;
	// Workgroups are dealt to the 8 XCDs round-robin (linear id mod 8), each with its own L2.  Groups that are neighbours in
	// depth-first order fill the same 128-byte lines of a row of F (their patch ids interleave), so XCD x takes the x-th
	// contiguous eighth of the groups: the partial lines of neighbouring groups meet in ONE L2 and leave it merged.
#if RTMI_FF_XCD_BLOCK
	// ... in BLOCKS of RTMI_FF_XCD_BLOCK groups dealt to the XCDs in turn (block b to XCD b mod 8): neighbouring groups of a block
	// still meet in one L2, and every XCD gets an even mix of the scene — with one contiguous eighth per XCD (round 2) a launch
	// whose rows lie in one wall left the XCD that owned that wall's columns (all zeros, no work) idle: 4096-row launches ran at
	// 85 G pair-samples/s where one launch of the whole matrix, balanced by symmetry, ran at 114 (profiles/ab_r03.md).
	const uint32_t xl = blockIdx.x >> 3;
	const uint32_t g = ((xl / RTMI_FF_XCD_BLOCK) * 8u + (blockIdx.x & 7u)) * RTMI_FF_XCD_BLOCK + (xl % RTMI_FF_XCD_BLOCK);
#else
	const uint32_t per_xcd = (gridDim.x + 7u) / 8u;
	const uint32_t g = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
#endif
	if (g >= n_groups) return; // the padded tail
	const uint32_t band = blockIdx.y * (kBlock / 64u) + uniform_u32(threadIdx.x >> 6);
	const uint32_t r_begin = band * band_rows, r_end = min(n_rows, r_begin + band_rows);
	if (r_begin >= n_rows) return; // whole wave
	const uint32_t j0 = T.dfs[(size_t)g * 64u]; // a group's first slot is always a patch
	const uint32_t j = T.dfs[(size_t)g * 64u + lane];
	const bool valid_j = j != RTMI_MISS;
	const PatchMeta m0 = T.meta[j0]; // wave-uniform
	// The patch a lane is evaluating: its own column — or, while the samples of the lanes that kept a sibling are dealt over the
	// wave (below), another lane's.  Everything about it that the sample loop reads:
	uint32_t jj, dj, leaf_j, behind_j;
	PatchMeta mj;
	const double *ejp;
	D3 nj;
	double area_term;
	const ChainEntry *chain_j;
	auto take_patch = [&](uint32_t jx) {
		jj = jx;
		mj = T.meta[jx];
		dj = meta_depth(mj);
		leaf_j = meta_leafpos(mj);
		ejp = P.verts + 9 * (size_t)jx;
		nj = d3(P.normals[3 * (size_t)jx], P.normals[3 * (size_t)jx + 1], P.normals[3 * (size_t)jx + 2]);
		area_term = P.area[jx] / (int)samples; // :319, the same for every sample
		chain_j = T.chain + mj.chain_off;
		behind_j = mj.behind & bits_below(dj); // which levels hold a sibling behind j's plane is j's own affair (PatchMeta::behind)
	};
	take_patch(valid_j ? j : j0);
	const ChainEntry *chain_0 = T.chain + m0.chain_off;
	// The group's paths agree on their first `common` levels: those siblings are the same for every lane (read once per wave
	// through the scalar cache, chain_0); below, each lane has its own.
	const uint32_t pdiff = mj.path ^ m0.path;
	const uint32_t common = wave_min_u32(min(dj, pdiff ? (uint32_t)(__ffs((int)pdiff) - 1) : 32u));
	const uint32_t low = bits_below(common);
	// Of the shared levels, the wave needs: those SOME lane has to test against its beam (not behind that lane's plane), and
	// those behind SOME lane's plane (a grazing ray tests them) — both the same for every row.
	const uint32_t cand_shared = wave_or_u32(~mj.behind & low), behind_shared = wave_or_u32(mj.behind & low);
	// the group's two constants of the "certain hit" test below (host: build_ff_tables), wave-uniform
	const uint4 gconst = t_gconst[g];
	const double sure_margin = (double)__uint_as_float(gconst.x), sure_cos = (double)__uint_as_float(gconst.y);
	// Same-plane rows (host: build_ff_tables): gkey != 0 = every patch of this group lies in ONE axis-parallel plane whose
	// coordinate is exact in every sample point (<= 20 significant bits: w·z + u·z + v·z = z with no rounding, w, u, v being
	// multiples of 2^-32 that sum to 1) and has an axis normal; a row i with the same key then has xj − xi exactly 0 along that
	// axis for every sample, hence cos_i = ±0, no sample counts (:320-324) and F(i, j) = 0 for the whole group — a sixth of a
	// box's matrix, ten samples each, without evaluating one of them.  (The root-traversal kernel and the oracle compute those
	// zeros; tests compare.)
	const uint32_t gkey = RTMI_FF_SAME_PLANE ? gconst.z : 0u;
	const uint32_t *t_pkey = reinterpret_cast<const uint32_t *>(t_gconst + n_groups);
	const unsigned long long m_flat = (planes_on != 0u && !(S.absmax > 1e4f)) ? ~0ull : 0ull; // the certain-hit test applies at all (see there)
	// dealing samples over the wave (below): lane L holds sample deal_k of the deal_q-th pair of a pass
	const uint32_t deal_per_pass = RTMI_FF_REGROUP && samples <= 64u ? 64u / samples : 0u;
	const uint32_t deal_m16 = uniform_u32((65536u + samples - 1u) / samples); // L div S = (L·m16) >> 16 for L < 64, S <= 64 (L·(m16·S − 2^16) < 2^16)
	for (uint32_t r = r_begin; r < r_end; ++r) {
		const uint32_t i = row0 + r; // wave-uniform
		double out = 0.0;
		if (i < P.n && !(gkey != 0u && t_pkey[i] == gkey)) {
			const PatchMeta mi = T.meta[i];
			const uint32_t di = meta_depth(mi), leaf_i = meta_leafpos(mi);
			const double *ei = P.verts + 9 * (size_t)i;
			const D3 ni = d3(P.normals[3 * (size_t)i], P.normals[3 * (size_t)i + 1], P.normals[3 * (size_t)i + 2]);
			const ChainEntry *chain_i = T.chain + mi.chain_off;
			const bool pair = valid_j && jj != i;
			// ---- the beam between the two patches against the siblings of both paths: once per pair ----------------
			// Every ray of the pair runs from a point within 1e-5 of patch i's bounds to its hit on triangle j, within
			// rounding of patch j's bounds; such a segment stays within max(half extents) of the line between the two
			// boxes' centres.  `margin` covers the 1e-5, the per-ray pad of the exact test and all fp32 roundings.
			// mask_i / mask_j: bit L = the sibling at level L of path i / path j may be met by a ray of this pair.
			// front_*: without the siblings that lie wholly behind (or in) the patch's own plane — flagged on the host, in
			// double, from exact bounds (build_ff_tables: nothing of the sibling is more than delta above the plane through the
			// patch's first vertex, heights taken along the caller's normal; the patch itself lies within delta/4 of it).  A ray
			// that leaves patch i with cos > graze_cos = 4e5·delta (host; 4e-6 in a room of size 10) is 1e-5·cos − delta/4 >
			// 3·delta above that plane at its origin and rises, so it meets such triangles only at t < 0; one that arrives on j
			// with cos > graze_cos is delta above j's plane no earlier than (delta + delta/4)/cos < 3.2e-6 before its hit on j,
			// so it meets them no earlier than that — where every hit gives the verdict the hit on j gives (|p − xj| < 1e-5
			// with a factor 3 to spare; the roundings of t and p are of relative size 1e-15/cos, a thousandth of that).
			// A flagged sibling costs NO beam test: it is in mask_* whatever the beam says (only a grazing ray looks at it, and
			// tests its box itself); the beam is tried on the others alone — in a room, the two or three top levels of a path.
			uint32_t front_i = 0u, front_j = 0u, um_i_front = 0u, um_j_front = 0u;
			const uint32_t xij = mi.path ^ mj.path, xi0 = mi.path ^ m0.path;
			const uint32_t lca = (pair && xij) ? (uint32_t)(__ffs((int)xij) - 1) : 0u;     // first level where the paths part
			const uint32_t lca0 = xi0 ? (uint32_t)(__ffs((int)xi0) - 1) : 32u;              // ... for the group's first patch (uniform)
			const uint32_t behind_i = mi.behind & bits_below(di);                            // uniform
			{
				float alo[3], ahi[3], blo[3], bhi[3];
				patch_bounds(ei, alo, ahi);
				// (patch j's bounds again for every row, through an index the compiler cannot match with `jj`: 9 loads and
				// ≈ 40 instructions per pair against six registers held over the whole kernel, one of them spilled)
				uint32_t jb = jj;
				asm volatile("" : "+v"(jb));
				patch_bounds(P.verts + 9 * (size_t)jb, blo, bhi);
				const float cax = 0.5f * (alo[0] + ahi[0]), cay = 0.5f * (alo[1] + ahi[1]), caz = 0.5f * (alo[2] + ahi[2]);
				const float cbx = 0.5f * (blo[0] + bhi[0]), cby = 0.5f * (blo[1] + bhi[1]), cbz = 0.5f * (blo[2] + bhi[2]);
				const float px = fmaxf(ahi[0] - alo[0], bhi[0] - blo[0]) * 0.5f + margin;
				const float py = fmaxf(ahi[1] - alo[1], bhi[1] - blo[1]) * 0.5f + margin;
				const float pz = fmaxf(ahi[2] - alo[2], bhi[2] - blo[2]) * 0.5f + margin;
				const RayS beam = make_rays(cax, cay, caz, cbx - cax, cby - cay, cbz - caz, px, py, pz);
				for (uint32_t m = ~mi.behind & bits_below(di); m; m &= m - 1u) { // wave-uniform entries: path i's unflagged levels
					FF_COUNT(81, 1);
					const uint32_t L = (uint32_t)__builtin_ctz(m);
					const float4 a = *reinterpret_cast<const float4 *>(chain_i + L), b = *(reinterpret_cast<const float4 *>(chain_i + L) + 1);
					const bool met = pair && L != lca && slab_box(a, b, beam, 1.0f) && beam_meets_leaf(S, __float_as_int(a.w), beam, stack);
					if (met) front_i |= 1u << L;
					if (__ballot(met) != 0ull) um_i_front |= 1u << L;
				}
				// path j above the group's common depth: if path i parts from the group there (lca0 < common), every lane's
				// lca is lca0 and the levels between are the group's shared siblings (their boxes: one scalar fetch per wave)
				for (uint32_t m = cand_shared & bits_above(lca0); m; m &= m - 1u) { // wave-uniform entries (none when lca0 >= common)
					FF_COUNT(82, 1);
					const uint32_t L = (uint32_t)__builtin_ctz(m);
					const float4 a = *reinterpret_cast<const float4 *>(chain_0 + L), b = *(reinterpret_cast<const float4 *>(chain_0 + L) + 1);
					const bool met = pair && !((behind_j >> L) & 1u) && slab_box(a, b, beam, 1.0f) && beam_meets_leaf(S, __float_as_int(a.w), beam, stack);
					if (met) front_j |= 1u << L;
					if (__ballot(met) != 0ull) um_j_front |= 1u << L;
				}
				for (uint32_t m = pair ? (~behind_j & bits_below(dj) & ~low & bits_above(lca)) : 0u; __ballot(m != 0u) != 0ull; m &= m - 1u) { // per-lane entries
					FF_COUNT(10, 1);
					if (m) {
						const uint32_t L = (uint32_t)__builtin_ctz(m);
						const float4 a = *reinterpret_cast<const float4 *>(chain_j + L), b = *(reinterpret_cast<const float4 *>(chain_j + L) + 1);
						if (slab_box(a, b, beam, 1.0f) && beam_meets_leaf(S, __float_as_int(a.w), beam, stack)) front_j |= 1u << L;
					}
				}
			}
			// what the WAVE loops over per sample (supersets of the lanes' own bits, which are tested per lane below)
			const uint32_t um_i_all = behind_i | um_i_front, um_j_all = (behind_shared & bits_above(lca0)) | um_j_front;
#ifdef RTMI_FF_COUNT
			{ // lanes whose pair keeps a sibling for its rays ("dirty"): how many per wave and row
				const uint32_t nd = (uint32_t)__popcll(__ballot(pair && (front_i | front_j) != 0u));
				FF_COUNT(83, 1); FF_COUNT(84, nd ? 1 : 0); FF_COUNT(85, nd);
				FF_COUNT(86, nd >= 1 && nd <= 6 ? 1 : 0); FF_COUNT(87, nd >= 7 && nd <= 12 ? 1 : 0); FF_COUNT(88, nd >= 13 && nd <= 25 ? 1 : 0); FF_COUNT(89, nd > 25 ? 1 : 0);
			}
#endif
			// ---- the samples of the pair, in the reference's order (:309-330) --------------------------------------
			// Lanes whose pair kept a sibling in front of a plane ("dirty": front_i | front_j != 0 — the same for all samples of the
			// pair) send EVERY sample down the long way: triangle j, the sibling boxes, perhaps a walk; and a wave takes the long
			// way for a sample as soon as one of its lanes does.  On the 64 K box 30 % of the (wave, row)s have such lanes, a
			// dozen of the 64 on average, and they cost the kernel two fifths of its time.  So those lanes sit out the sample loop
			// (the others then take the certain-hit path every time), and afterwards THEIR samples are dealt over the wave:
			// lane L takes sample L mod S of the (L div S)-th such pair — 64 div S pairs per pass, all their samples at once.
			// The sums stay the reference's: a pair's terms reach its own lane through ds_bpermute and are added in sample order.
			// Not worth it when the passes would be more than half as many as the samples (then every lane stays in the loop).
			// (Only where the other lanes can take the certain path at all: flat patches with geometric normals, coordinates below 1e4.)
			const bool dirty = RTMI_FF_REGROUP && RTMI_FF_SKIP_MT && planes_on != 0u && !(S.absmax > 1e4f) && pair && (front_i | front_j) != 0u;
			const unsigned long long dmask = __ballot(dirty);
			const uint32_t n_dirty = (uint32_t)__popcll(dmask);
			const uint32_t passes_wanted = deal_per_pass ? (n_dirty + deal_per_pass - 1u) / deal_per_pass : 0u;
			const uint32_t n_pass = passes_wanted * RTMI_FF_REGROUP_DEN <= samples * RTMI_FF_REGROUP_NUM ? passes_wanted : 0u;
			const uint32_t my_rank = (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(dmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)dmask, 0u)); // dirty lanes below this one
			unsigned long long undealt = dmask;
			// ---- the samples of the pair, in the reference's order (:309-330) --------------------------------------
			uint64_t h0 = mix64(seed ^ (((uint64_t)i << 32) | jj));
			uint32_t e_front_i = front_i, e_front_j = front_j, e_lca = lca; // of the pair the lane is evaluating
			unsigned long long m_act = __ballot(pair && !(n_pass != 0u && dirty));  // lanes that evaluate a pair in this iteration
			unsigned long long m_clean = __ballot((front_i | front_j) == 0u);        // ... whose pair kept no sibling in front of a plane
			double ff = 0.0;
			for (uint32_t t = 0; t < samples + n_pass; ++t) {
				uint32_t k = t;
				bool act = pair && !(n_pass != 0u && dirty);
				if (t >= samples) { // a pass over the dealt samples (wave-uniform branch)
					FF_COUNT(90, 1);
					const uint32_t deal_q = (lane * deal_m16) >> 16, deal_k = lane - deal_q * samples;
					uint32_t src = 0u;
					act = false;
					for (uint32_t q = 0; q < deal_per_pass && undealt != 0ull; ++q) {
						const uint32_t b = (uint32_t)__ffsll((long long)undealt) - 1u;
						undealt &= undealt - 1ull;
						if (deal_q == q) { src = b; act = true; }
					}
					const uint32_t jx = (uint32_t)__shfl((int)j, (int)src, 64);
					e_front_i = (uint32_t)__shfl((int)front_i, (int)src, 64);
					e_front_j = (uint32_t)__shfl((int)front_j, (int)src, 64);
					take_patch(act ? jx : (valid_j ? j : j0)); // (a lane without a share evaluates its own column, masked)
					const uint32_t x = mi.path ^ mj.path;
					e_lca = x ? (uint32_t)(__ffs((int)x) - 1) : 0u;
					h0 = mix64(seed ^ (((uint64_t)i << 32) | jj));
					k = deal_k;
					m_act = __ballot(act);
					m_clean = __ballot((e_front_i | e_front_j) == 0u);
				}
				double add = 0.0; // what this sample adds to its pair's sum
				do {
				// Every lane evaluates its sample, pair or not (the diagonal lane and the padded tail read patch jj like any other and
				// are masked by `pair`): no branch around the arithmetic, and no values to preset for the lanes that would skip it.
				const D3 xi = sample_point_h(ei, h0, k, 0);
				double bmin_j;
				const D3 xj = sample_point_hb(ejp, h0, k, 1, bmin_j);
				// Unit(xj − xi) once: the reference evaluates it three times (:235, :317, and Unit(xi − xj) = its exact
				// negation, :318), and dist2 = (xi−xj)·(xi−xj) is the squared length inside it — same IEEE operations.
				const D3 vec = xj - xi;
				const double dist2 = len2(vec);
				// (The reference's Unit() returns 0 for a zero vector, vec3.h:98-101.  Here xj == xi gives 0/0: every component a
				// NaN, both cosines NaNs, `shoot` false — what the zero vector gives (cosines ±0), without the branch around the
				// root and the division.  A NaN direction is never used: everything below is under `shoot` or one of its subsets.)
				const D3 dir = unit_of_len2(vec, dist2);
				// A sample adds to the sum only if delF > 0, i.e. both cosines are positive (:320-324) — whatever the
				// visibility ray says.  So the ray is shot only for those samples.
				const double c1 = dot(dir, ni);
				const double c2 = -dot(dir, nj);
				const D3 org = xi + (kEpsilon * dir);
				// (votes as ANDs of the votes on single comparisons: a vote on a comparison is the comparison's own lane mask, a vote on
				// an AND of them costs two more vector instructions)
				const bool pos1 = c1 > 0.0, pos2 = c2 > 0.0;
				const bool shoot = act & pos1 & pos2;
				const unsigned long long m_shoot = __ballot(pos1) & __ballot(pos2) & m_act;
				FF_COUNT(0, 1);
				if (m_shoot == 0ull) break;
				FF_COUNT(13, 1); FF_COUNT_LANES(9, shoot);
				// Triangle i lies 1e-5 BEHIND the origin along the ray (t = −1e-5 up to rounding of relative size 1e-15/c1): it
				// cannot be hit unless the ray all but grazes it — then it is tested like any other, and so is everything
				// behind the two planes.
				// (each end on its own: what lies behind i's plane matters to a ray that grazes i, whatever it does at j)
				const bool graze_i = !(c1 > graze_cos_i), graze_j = !(c2 > graze_cos_j);
#if RTMI_FF_SKIP_MT
				// ---- the certain hit: nothing to test at all --------------------------------------------------------------
				// A ray aimed at a point WELL INSIDE triangle j, not grazing either plane, with no sibling of either path left
				// by the beam for it (s_i = s_j = 0): the exact test on j would accept it, nothing else can stop it, and the
				// verdict of the hit on j is "visible" (the argument above).  The wave skips the triangle test on j and the
				// walk when EVERY lane that shoots is such a lane.  "Would accept": the ray passes within 3e-15·|coord|max of
				// xj = w·v1 + u·v2 + v·v3 (roundings of xj, dir, org), so its exact crossing of j's plane has barycentrics within
				// 6e-15·|coord|max/(c2·a) of (u, v), a = j's smallest altitude; the test's own u', v' (object.cpp:46-56) differ
				// from those by the roundings of tvec·pvec, dir·qvec and det = ±2A·c2 — below 2e-15·(|coord|max + edge)·edge/(2A·c2).
				// The host's per-group constant sure_margin = 1e-12·(1 + |coord|max)·max over the group of (1 + edge)/a is a hundred
				// times their sum: bmin·c2 > sure_margin puts u', v', u' + v' strictly inside [0, 1]; c2 > sure_cos >= 1.01e-8/(2A)
				// keeps |det| out of the ±1e-8 window (and is >= graze_cos_j); dist² > 1e-6 keeps t = dist − 1e-5 (error below
				// 1e-3·sure_margin/c2 < 3.4e-4) positive.  Flat, geometric-normal patches only (planes_on, as for the flags).
				{
					const unsigned long long m_sure = __ballot(c1 > graze_cos_i) & __ballot(c2 > sure_cos) & __ballot(bmin_j * c2 > sure_margin) &
					                                  __ballot(dist2 > 1e-6) & m_clean & m_flat;
#ifdef RTMI_FF_TIMING_ALL_SURE // timing experiment only (wrong results wherever something is in the way): every sample takes the certain path
					if (true) {
#else
					if ((m_shoot & ~m_sure) == 0ull) {
#endif
						FF_COUNT(79, 1);
						if (shoot) {
							const double cosFactor = c1 * c2;
							const double delF = cosFactor / ((kPi * dist2) + area_term);
							if (delF > 0.0) add = delF;
						}
						break;
					}
				}
#endif
				double best = RTMI_INFINITY;
				uint32_t pos = RTMI_MISS;
				int cur = kSentinel, sp = 0;
				bool walk = false;     // this lane's ray is bounded by its hit on triangle j: siblings of the two paths
				bool from_root = false; // ... or not (or too many siblings are hit): the ordinary traversal
				if (shoot) {
					// triangle j first: e1 = v2 − v1, e2 = v3 − v1 are the IEEE subtractions the scene stores (object.cpp:36-37)
					GTri Tj;
#pragma unroll
					for (int a = 0; a < 3; ++a) { Tj.v0[a] = ejp[a]; Tj.e1[a] = ejp[3 + a] - ejp[a]; Tj.e2[a] = ejp[6 + a] - ejp[a]; }
					double u = 0.0, v = 0.0;
					const double t = mt_exact(org.x, org.y, org.z, dir.x, dir.y, dir.z, Tj, u, v);
					if (t > 0.0) { best = t; pos = leaf_j; walk = true; }
					else from_root = true;
				}
				auto push = [&](int ref) {
					if ((uint32_t)sp < pend_cap) { stack[sp * kBlock] = ref; ++sp; }
					else from_root = true;
				};
				if (walk && (graze_i || S.absmax > 1e4f)) push((int)~(leaf_i << 3));
				// mask_i / mask_j: every sibling a grazing ray has to look at — those behind the plane as well (not the one the other
				// patch lies in: level lca of path i; nor path j's above the parting)
				const uint32_t s_i = graze_i ? ((behind_i & ~(1u << e_lca)) | e_front_i) : e_front_i;
				const uint32_t s_j = graze_j ? ((behind_j & bits_above(e_lca)) | e_front_j) : e_front_j;
				const bool any_graze_i = __ballot(walk && graze_i) != 0ull, any_graze_j = __ballot(walk && graze_j) != 0ull;
				const bool any_graze = any_graze_i || any_graze_j; (void)any_graze;
				const uint32_t um_i = any_graze_i ? um_i_all : um_i_front, um_j = any_graze_j ? um_j_all : um_j_front;
				const uint32_t own_j = s_j & ~low;
				// the siblings the beam met, against THIS ray (exact slab test, bounded by the hit on j) — two waves in three have
				// none for a sample, and skip the fp32 ray with them
				if ((um_i | um_j) != 0u || __ballot(walk && own_j != 0u) != 0ull) {
				const float tb0 = f32_above(best);
				RayS rs;
				{
					const float fx = (float)org.x, fy = (float)org.y, fz = (float)org.z;
					const float pad = 3.814697265625e-06f * (S.absmax + fmaxf(fabsf(fx), fmaxf(fabsf(fy), fabsf(fz)))) + 1e-30f; // as make_rayf
					rs = make_rays(fx, fy, fz, (float)dir.x, (float)dir.y, (float)dir.z, pad, pad, pad);
				}
				for (uint32_t m = um_i; m; m &= m - 1u) { // wave-uniform
					FF_COUNT(1, 1);
					const uint32_t L = (uint32_t)__builtin_ctz(m);
					const float4 a = *reinterpret_cast<const float4 *>(chain_i + L), b = *(reinterpret_cast<const float4 *>(chain_i + L) + 1);
					if (walk && ((s_i >> L) & 1u) && slab_box(a, b, rs, tb0)) { push(__float_as_int(RTMI_FF_WIDE_WALK ? b.w : a.w)); FF_COUNT(16 + L, 1); }
				}
				for (uint32_t m = um_j; m; m &= m - 1u) { // wave-uniform
					FF_COUNT(2, 1);
					const uint32_t L = (uint32_t)__builtin_ctz(m);
					const float4 a = *reinterpret_cast<const float4 *>(chain_0 + L), b = *(reinterpret_cast<const float4 *>(chain_0 + L) + 1);
					if (walk && ((s_j >> L) & 1u) && slab_box(a, b, rs, tb0)) { push(__float_as_int(RTMI_FF_WIDE_WALK ? b.w : a.w)); FF_COUNT(48 + L, 1); }
				}
				for (uint32_t m = walk ? own_j : 0u; __ballot(m != 0u) != 0ull; m &= m - 1u) { // per lane
					FF_COUNT(3, 1);
					if (m) {
						const uint32_t L = (uint32_t)__builtin_ctz(m);
						const float4 a = *reinterpret_cast<const float4 *>(chain_j + L), b = *(reinterpret_cast<const float4 *>(chain_j + L) + 1);
						if (slab_box(a, b, rs, tb0)) { push(__float_as_int(RTMI_FF_WIDE_WALK ? b.w : a.w)); FF_COUNT(48 + L, 1); }
					}
				}
				}
				FF_COUNT_LANES(8, from_root); FF_COUNT(4, __popcll(__ballot(sp > 0))); FF_COUNT(14, __popcll(__ballot(sp > 1))); FF_COUNT(15, __popcll(__ballot(sp > 2)));
				if (from_root) { best = RTMI_INFINITY; pos = RTMI_MISS; sp = 0; cur = S.root_ref; }
				else if (sp > 0) { --sp; cur = stack[sp * kBlock]; }
				if (__ballot(cur != kSentinel) != 0ull) {
					FF_COUNT(5, 1); FF_COUNT(72, any_graze ? 1 : 0); FF_COUNT_LANES(73, walk && (graze_i || graze_j) && cur != kSentinel); FF_COUNT_LANES(78, cur != kSentinel);
#if RTMI_FF_WIDE_WALK
					const RayW rf = make_rayw(S, org.x, org.y, org.z, dir.x, dir.y, dir.z);
#else
					const RayF rf = make_rayf(S, org.x, org.y, org.z, dir.x, dir.y, dir.z);
#endif
					trace_from(S, rf, org.x, org.y, org.z, dir.x, dir.y, dir.z, best, pos, cur, sp, stack, any_graze ? 1u : 0u);
				}
				if (shoot && pos != RTMI_MISS) {
					// hit.point = o + (d·((v1 − o)·n)) / (n·d) with the HIT triangle's first vertex and normal (object.cpp:23),
					// visible ⇔ |hit.point − xj| < 1e-5 ∧ d·n_j < 0 (radiosity.cpp:249-252; the latter is c2 > 0)
					// (nearly always the hit IS triangle j: its first vertex and normal are at hand — the same bits as the two
					// dependent loads through tris[pos] would fetch, which only a ray stopped by something else needs)
					// When the hit IS triangle j, the planes are known to be planes (planes_on: every caller normal is the
					// geometric one and each patch lies within delta/4 of the plane through its first vertex along it — the host's
					// premise for the flags above) and the ray does not graze j (c2 > graze_cos_j = 4e5·delta), the verdict is known
					// without p: the ray's line passes within 1e-14·|coord| of xj (roundings of dir and org), xj lies within delta/4
					// of j's plane, so the line meets that plane within (delta/4 + 1e-14·|coord|)/c2 < 6.3e-7 of xj; the computed p
					// differs from that point by the roundings of (v1 − o)·n and n·d divided by c2, < 3e-15·|coord|/c2 < 1e-8·|coord|max
					// /(1 + |coord|max).  Together < 7e-7 against the 1e-5 of the test: visible, as the full evaluation finds
					// (tests compare the two kernels and the oracle bit for bit; RTMI_FF_SKIP_P=0 builds the full evaluation).
#if RTMI_FF_SKIP_P
					bool vis = planes_on != 0u && pos == leaf_j && !graze_j;
#else
					bool vis = false;
#endif
					if (!vis) {
						D3 n = nj, v1 = d3(ejp[0], ejp[1], ejp[2]);
						if (pos != leaf_j) {
							const GTri &Th = S.tris[pos];
							const double *np = P.normals + 3 * (size_t)Th.orig;
							n = d3(np[0], np[1], np[2]);
							v1 = d3(Th.v0[0], Th.v0[1], Th.v0[2]);
						}
						const D3 p = org + (dot(v1 - org, n) * dir) / dot(n, dir);
						vis = len2(p - xj) < kEpsilonSquaredExact; // sqrt(...) < 1e-5 exactly (rtmi_rad_device.hpp)
					}
					if (vis) {
						const double cosFactor = c1 * c2;
						const double delF = cosFactor / ((kPi * dist2) + area_term);
						if (delF > 0.0) add = delF;
					}
				}
				} while (0);
				if (t < samples) ff += add; // (a sum that starts at +0 and takes positive terms is never −0: adding +0 changes nothing)
				else {
					// the pass's terms go home: a dealt pair's samples sit in S consecutive lanes, its own lane adds them in order
					const uint32_t first = my_rank - (t - samples) * deal_per_pass; // my pair's place in this pass, if it is in it
					const bool mine = dirty && first < deal_per_pass;
					for (uint32_t kk = 0; kk < samples; ++kk) {
						const double term = shfl_f64(add, (mine ? first : 0u) * samples + kk);
						if (mine) ff += term;
					}
				}
			}
			if (n_pass != 0u) { // back to the lane's own column (through an index the compiler cannot match with the first call's: the
				// addresses it would otherwise keep over the whole kernel for this were spilled)
				uint32_t jo = valid_j ? j : j0;
				asm volatile("" : "+v"(jo));
				take_patch(jo);
			}
			uint32_t ns = samples; // (converted here, per row: held as a double over the kernel it was spilled)
			asm volatile("" : "+s"(ns));
			ff = ff / ((double)ns);
			// (area_j read again, through an index the compiler cannot match with line 269's: kept in registers over the row
			// it was the one value the kernel spilled.)
			uint32_t jr = jj;
			asm volatile("" : "+v"(jr));
			out = ff * P.area[jr];
		}
		if (valid_j) {
			uint32_t js = j; // (likewise: the hoisted address F + 4 j was spilled; the index costs one register, not two)
			asm volatile("" : "+v"(js));
			F[(uint64_t)r * ld + js] = (float)out;
		}
	}
}

#ifndef RTMI_FF_FLAT_WAVES
#define RTMI_FF_FLAT_WAVES 7
#endif
// ---- the round-1 kernel, kept as a cross-check (RTMI_FF_KERNEL=flat; tests compare the two) -----------------------------
// One lane per matrix entry (i, j), j fastest; every ray descends from the root.
RTMI_TRAVERSAL_KERNEL_W(RTMI_FF_FLAT_WAVES) void k_form_factors_flat(SceneDev S, PatchesDev P, uint32_t samples, uint64_t seed, uint32_t row0,
                                                    uint32_t n_rows, float *__restrict__ F, uint64_t ld) {
	extern __shared__ int stack_all[];
	int *stack = stack_all + threadIdx.x;
	const uint64_t idx = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
	if (idx >= (uint64_t)n_rows * P.n) return;
	const uint32_t r = (uint32_t)(idx / P.n), j = (uint32_t)(idx % P.n), i = row0 + r;
	double out = 0.0;
	if (i < P.n && i != j) {
		const double *ei = P.verts + 9 * (size_t)i, *ejp = P.verts + 9 * (size_t)j;
		const D3 ni = d3(P.normals[3 * (size_t)i], P.normals[3 * (size_t)i + 1], P.normals[3 * (size_t)i + 2]);
		const D3 nj = d3(P.normals[3 * (size_t)j], P.normals[3 * (size_t)j + 1], P.normals[3 * (size_t)j + 2]);
		const double area_j = P.area[j];
		const double area_term = area_j / (int)samples;
		double ff = 0.0;
		for (uint32_t k = 0; k < samples; ++k) {
			const D3 xi = sample_point(ei, seed, i, j, k, 0), xj = sample_point(ejp, seed, i, j, k, 1);
			const D3 vec = xj - xi;
			const double dist2 = len2(vec);
			const D3 dir = dist2 > 0.0 ? vec / sqrt(dist2) : d3(0, 0, 0);
			const double c1 = dot(dir, ni);
			const double c2 = -dot(dir, nj);
			if (c1 > 0.0 && c2 > 0.0 && visible_along(S, P, j, xi + (kEpsilon * dir), dir, xj, stack)) {
				const double cosFactor = c1 * c2;
				const double delF = cosFactor / ((kPi * dist2) + area_term);
				if (delF > 0.0) ff += delF;
			}
		}
		ff = ff / ((double)samples);
		out = ff * area_j;
	}
	F[(uint64_t)r * ld + j] = (float)out;
}

// ---- host: the path tables ---------------------------------------------------------------------------------------------
static float f32_below(double x) { float f = (float)x; return (double)f > x ? std::nextafterf(f, -INFINITY) : f; }
static float f32_above_h(double x) { float f = (float)x; return (double)f < x ? std::nextafterf(f, INFINITY) : f; }

// Walks the binary hierarchy as the device holds it and writes, per patch, the siblings along its path.
struct Box3 { double lo[3], hi[3]; };
struct TableWalk {
	const rtmi_scene *s;
	const double *verts = nullptr, *normals = nullptr; // host copies, original order
	bool flags_on = false;  // the caller's normals are the triangles' geometric normals and the scene is small enough in coordinates
	double delta = 0.0;
	std::vector<Box3> sib_box; // exact (double) bounds of sib[L]'s subtree
	std::vector<QNode> nodes;
	std::vector<GTri> tris;
	std::vector<ChainEntry> chain, sib; // sib[L] = the sibling at level L of the path being walked
	std::vector<PatchMeta> meta;
	std::vector<uint32_t> dfs;
	std::vector<uint8_t> seen;
	bool bad = false;
	double plane(uint32_t q, int a) const { return (double)s->dev.qlo[a] + (double)q * (double)s->dev.qstep[a]; }
	// exact bounds of every subtree, bottom-up (index: inner node id; leaves are looked up directly)
	std::vector<Box3> node_box;
	Box3 leaf_box(int32_t ref) const {
		const uint32_t id = tris[(~(uint32_t)ref) >> 3].orig;
		const double *v = verts + 9 * (size_t)id;
		Box3 b;
		for (int a = 0; a < 3; ++a) {
			b.lo[a] = std::min(v[a], std::min(v[3 + a], v[6 + a]));
			b.hi[a] = std::max(v[a], std::max(v[3 + a], v[6 + a]));
		}
		return b;
	}
	Box3 bounds(int32_t ref, uint32_t depth) {
		if (bad) return Box3();
		if (ref < 0) {
			const uint32_t code = ~(uint32_t)ref, pos = code >> 3;
			if ((code & 7u) != 0u || pos >= tris.size() || tris[pos].orig >= meta.size()) { bad = true; return Box3(); }
			return leaf_box(ref);
		}
		if ((size_t)ref >= nodes.size() || depth >= 31u) { bad = true; return Box3(); }
		const QNode &q = nodes[ref];
		const Box3 l = bounds((int32_t)q.b.z, depth + 1u), r = bounds((int32_t)q.b.w, depth + 1u);
		Box3 b;
		for (int a = 0; a < 3; ++a) { b.lo[a] = std::min(l.lo[a], r.lo[a]); b.hi[a] = std::max(l.hi[a], r.hi[a]); }
		node_box[ref] = b;
		return b;
	}
	Box3 box_of(int32_t ref) const { return ref < 0 ? leaf_box(ref) : node_box[ref]; }
	void walk(int32_t ref, uint32_t path, uint32_t depth) {
		if (bad) return;
		if (ref < 0) { // a leaf: ONE triangle (rtmi_scene_create builds leaves of one)
			const uint32_t code = ~(uint32_t)ref, pos = code >> 3;
			if ((code & 7u) != 0u || pos >= tris.size() || pos >= kMaxFirst) { bad = true; return; }
			const uint32_t id = tris[pos].orig;
			if (id >= meta.size() || seen[id]) { bad = true; return; }
			seen[id] = 1;
			meta[id].chain_off = (uint32_t)chain.size();
			meta[id].path = path;
			meta[id].depth_leafpos = (depth << 27) | pos; // depth < 32 (checked below), pos < kMaxFirst = 2^27 (leaf references)
			meta[id].behind = 0u;
			chain.insert(chain.end(), sib.begin(), sib.end());
			if (flags_on) { // which siblings lie wholly behind (or in) this patch's plane: support of their exact bounds along the normal
				const double *v1 = verts + 9 * (size_t)id, *nr = normals + 3 * (size_t)id;
				if (normal_is_geometric(id))
					for (uint32_t L = 0; L < depth; ++L) {
						const Box3 &b = sib_box[L];
						double sup = 0.0;
						for (int a = 0; a < 3; ++a) sup += std::max(nr[a] * (b.lo[a] - v1[a]), nr[a] * (b.hi[a] - v1[a]));
						if (sup <= delta) meta[id].behind |= 1u << L;
					}
			}
			dfs.push_back(id);
			return;
		}
		if ((size_t)ref >= nodes.size() || depth >= 31u) { bad = true; return; }
		const QNode &q = nodes[ref];
		const uint32_t w[2][3] = {{q.a.x, q.a.y, q.a.z}, {q.a.w, q.b.x, q.b.y}};
		const int32_t child[2] = {(int32_t)q.b.z, (int32_t)q.b.w};
		ChainEntry e[2];
		for (int c = 0; c < 2; ++c) {
			for (int a = 0; a < 3; ++a) { // the box the traversal kernels test for this child, as fp32 planes that contain it
				e[c].lo[a] = f32_below(plane(w[c][a] & 0xffffu, a));
				e[c].hi[a] = f32_above_h(plane(w[c][a] >> 16, a));
			}
			e[c].ref = child[c];
			e[c].wide = child[c];
#if RTMI_FF_WIDE_WALK
			if (child[c] >= 0) {
				if ((size_t)child[c] >= s->wide_of_node.size() || s->wide_of_node[child[c]] == 0xffffffffu) { bad = true; return; }
				e[c].wide = (int32_t)s->wide_of_node[child[c]];
			}
#endif
		}
		sib.push_back(e[1]); // walking into child 0, the other child is the sibling at this level
		sib_box.push_back(flags_on ? box_of(child[1]) : Box3());
		walk(child[0], path, depth + 1u);
		sib.back() = e[0];
		sib_box.back() = flags_on ? box_of(child[0]) : Box3();
		walk(child[1], path | (1u << depth), depth + 1u);
		sib.pop_back();
		sib_box.pop_back();
	}
	// the caller's normal of patch id equals ± Unit((v2 − v1) ^ (v3 − v1)) to 1e-9 per component, AND the patch lies in the
	// plane through its first vertex perpendicular to that normal to delta/4 (its other two vertices do, hence all of it):
	// heights along the CALLER's normal are what the flags below and the kernel's cosines measure, so this is the bound on
	// how far a point of the patch itself — a ray's origin on i, its hit on j — can sit off the plane the flags refer to
	bool normal_is_geometric(uint32_t id) const {
		const double *v = verts + 9 * (size_t)id, *nr = normals + 3 * (size_t)id;
		const double ax = v[3] - v[0], ay = v[4] - v[1], az = v[5] - v[2], bx = v[6] - v[0], by = v[7] - v[1], bz = v[8] - v[2];
		double g[3] = {ay * bz - az * by, az * bx - ax * bz, ax * by - ay * bx};
		const double len = std::sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
		if (!(len > 0.0)) return false;
		if (std::fabs(nr[0] * ax + nr[1] * ay + nr[2] * az) > 0.25 * delta || std::fabs(nr[0] * bx + nr[1] * by + nr[2] * bz) > 0.25 * delta) return false;
		double dp = 0.0, dm = 0.0;
		for (int a = 0; a < 3; ++a) {
			dp = std::max(dp, std::fabs(nr[a] - g[a] / len));
			dm = std::max(dm, std::fabs(nr[a] + g[a] / len));
		}
		return std::min(dp, dm) <= 1e-9;
	}
	// Degenerate patches (zero area) are never hit (|det| < 1e-8): they need no normal.
	bool all_normals_geometric() const {
		for (uint32_t id = 0; id < (uint32_t)meta.size(); ++id) {
			const double *v = verts + 9 * (size_t)id;
			const double ax = v[3] - v[0], ay = v[4] - v[1], az = v[5] - v[2], bx = v[6] - v[0], by = v[7] - v[1], bz = v[8] - v[2];
			const double gx = ay * bz - az * by, gy = az * bx - ax * bz, gz = ax * by - ay * bx;
			if (gx * gx + gy * gy + gz * gz > 0.0 && !normal_is_geometric(id)) return false;
		}
		return true;
	}
};

static int build_ff_tables(rtmi_patches *P, const char *who) {
	std::lock_guard<std::mutex> lock(P->ff_mutex);
	if (P->ff_ready) return 0;
	const auto t_begin = std::chrono::steady_clock::now();
	auto ms_since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
	const rtmi_scene *s = P->scene;
	const uint32_t n = P->n;
	TableWalk W;
	W.s = s;
	W.nodes.resize((size_t)s->info.n_nodes);
	W.tris.resize(n);
	W.meta.resize(n);
	W.seen.assign(n, 0);
	if ((!W.nodes.empty() && hipMemcpy(W.nodes.data(), s->dev.nodes, W.nodes.size() * sizeof(QNode), hipMemcpyDeviceToHost) != hipSuccess) ||
	    hipMemcpy(W.tris.data(), s->dev.tris, (size_t)n * sizeof(GTri), hipMemcpyDeviceToHost) != hipSuccess)
		return fail(std::string(who) + ": reading the hierarchy back failed");
	std::vector<double> h_verts(9 * (size_t)n), h_normals(3 * (size_t)n);
	if (hipMemcpy(h_verts.data(), P->d_verts, h_verts.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess ||
	    hipMemcpy(h_normals.data(), P->d_normals, h_normals.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
		return fail(std::string(who) + ": reading the patches back failed");
	W.verts = h_verts.data();
	W.normals = h_normals.data();
	// "behind its patch's plane" flags: only when the p formula of ANY triangle that might win a tie is the true plane
	// intersection (geometric normals throughout), and while delta is small enough for the kernel's cosine threshold
	// 4e5·delta (graze_cos below) to stay at 5e-4 or less
	W.delta = 9.094947017729282e-13 * (1.0 + (double)s->dev.absmax); // 2^-40·(1 + |coord|max)
	const char *planes = std::getenv("RTMI_FF_PLANES"); // "0": no plane flags (tests compare both settings)
	W.flags_on = W.delta <= 1.25e-9 && !(planes && planes[0] == '0') && W.all_normals_geometric();
	const double ms_read = ms_since();
	W.node_box.resize(W.nodes.size());
	if (W.flags_on) (void)W.bounds(s->dev.root_ref, 0u);
	W.dfs.reserve(((size_t)n + 63) / 64 * 64);
	W.chain.reserve((size_t)n * std::min<size_t>(s->info.max_depth, 24)); // (an estimate: a balanced tree's paths are log2 n long; no reallocation on the way for those)
	W.walk(s->dev.root_ref, 0u, 0u);
	const double ms_walk = ms_since();
	if (W.bad || W.dfs.size() != n) return fail(std::string(who) + ": the hierarchy is not a binary tree over the patches with one triangle per leaf");
	if (W.chain.size() > 0xffffffffull) return fail(std::string(who) + ": path table too large");
	const uint32_t n_groups = (n + 63u) / 64u;
	{
		const char *ord = std::getenv("RTMI_FF_ORDER"); // "id": groups of 64 consecutive patch ids (experiment)
		if (ord && std::strcmp(ord, "id") == 0)
			for (uint32_t k = 0; k < n; ++k) W.dfs[k] = k;
	}
	W.dfs.resize((size_t)n_groups * 64u, RTMI_MISS);
	if (W.chain.empty()) W.chain.emplace_back(); // a one-patch scene has no siblings; keep the pointer valid
	// The cosines below which a ray "grazes" a patch's plane and tests the flagged siblings after all (k_form_factors).  They
	// scale with delta: 1.25·delta/graze_cos_j = 3.1e-6 is the most a flagged triangle can come before the hit on j.  (Fixed at
	// 1e-3 — the value the largest admissible delta needs — until round 2's last day: in a room of size 10 that made 0.1 % of
	// the rays graze, and those, running along a wall inside the pad of dozens of its leaves, cost 15-30 % of the kernel.)
	// At the leaving end the ray only has to START above the flagged siblings: 1e-5·cos > 4·delta (the roundings of the
	// origin and of the host's flags are 1e-4 of delta).
	const double graze_cos = W.flags_on ? 4.0e5 * W.delta : 1e-3;
	// Per group of 64 patches, the two constants of the kernel's "certain hit" test (see there): with a = 2A/edge the
	// smallest altitude and edge the longest edge of a patch,
	//   sure_margin = 1e-12·(1 + |coord|max)·max (1 + edge)/a     sure_cos = max(graze_cos, max 1.01e-8/(2A))
	// rounded up to float; a patch without area makes both infinite (its group then always runs the exact test).
	// Same-plane keys (k_form_factors): patch id gets key k >= 1 iff its caller normal is exactly ± a coordinate axis, its three
	// vertices have the very same coordinate z along that axis, and z has at most 20 significant bits (every sample point of the
	// patch then has exactly z there); the key numbers the (axis, z) pairs.  A group's key is its patches' common key, else 0.
	std::vector<uint32_t> pkey(n, 0u);
	{
		std::unordered_map<uint64_t, uint32_t> ids[3];
		for (uint32_t id = 0; id < n; ++id) {
			const double *v = h_verts.data() + 9 * (size_t)id, *nr = h_normals.data() + 3 * (size_t)id;
			for (int a = 0; a < 3; ++a) {
				const int b = (a + 1) % 3, c = (a + 2) % 3;
				if (!(std::fabs(nr[a]) == 1.0 && nr[b] == 0.0 && nr[c] == 0.0)) continue;
				const double z = v[a] + 0.0;
				if (!(v[3 + a] == z && v[6 + a] == z) || !std::isfinite(z)) continue;
				int e = 0;
				const double m = std::frexp(z, &e) * 1048576.0; // 2^20
				if (m != std::floor(m)) continue;
				uint64_t bits;
				std::memcpy(&bits, &z, sizeof bits);
				auto it = ids[a].find(bits);
				if (it == ids[a].end()) it = ids[a].emplace(bits, (uint32_t)(ids[0].size() + ids[1].size() + ids[2].size() + 1)).first;
				pkey[id] = it->second;
			}
		}
	}
	std::vector<uint32_t> gconst(4 * (size_t)n_groups + n, 0u);
	auto f32_bits = [](float f) { uint32_t u; std::memcpy(&u, &f, sizeof u); return u; };
	for (uint32_t id = 0; id < n; ++id) gconst[4 * (size_t)n_groups + id] = pkey[id];
	for (uint32_t g = 0; g < n_groups; ++g) {
		double margin = 0.0, cosmin = graze_cos;
		for (uint32_t k = 0; k < 64u; ++k) {
			const uint32_t id = W.dfs[(size_t)g * 64u + k];
			if (id == RTMI_MISS) continue;
			const double *v = h_verts.data() + 9 * (size_t)id;
			const double a[3] = {v[3] - v[0], v[4] - v[1], v[5] - v[2]}, b[3] = {v[6] - v[0], v[7] - v[1], v[8] - v[2]}, c[3] = {v[6] - v[3], v[7] - v[4], v[8] - v[5]};
			const double cr[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
			const double twoA = std::sqrt(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]);
			const double edge = std::sqrt(std::max(a[0] * a[0] + a[1] * a[1] + a[2] * a[2], std::max(b[0] * b[0] + b[1] * b[1] + b[2] * b[2], c[0] * c[0] + c[1] * c[1] + c[2] * c[2])));
			if (!(twoA > 0.0) || !std::isfinite(twoA) || !std::isfinite(edge)) { margin = cosmin = INFINITY; break; }
			margin = std::max(margin, 1e-12 * (1.0 + (double)s->dev.absmax) * (1.0 + edge) * edge / twoA);
			cosmin = std::max(cosmin, 1.01e-8 / twoA);
		}
		gconst[4 * (size_t)g] = f32_bits(f32_above_h(margin));
		gconst[4 * (size_t)g + 1] = f32_bits(f32_above_h(cosmin));
		uint32_t key = 0u;
		bool first = true;
		for (uint32_t k = 0; k < 64u; ++k) {
			const uint32_t id = W.dfs[(size_t)g * 64u + k];
			if (id == RTMI_MISS) continue;
			if (first) { key = pkey[id]; first = false; }
			else if (pkey[id] != key) key = 0u;
		}
		gconst[4 * (size_t)g + 2] = key;
	}
	const double ms_consts = ms_since();
	void *d_chain = nullptr, *d_meta = nullptr, *d_gconst = nullptr;
	uint32_t *d_dfs = nullptr;
	bool ok = hipMalloc(&d_chain, W.chain.size() * sizeof(ChainEntry)) == hipSuccess && hipMalloc(&d_meta, (size_t)n * sizeof(PatchMeta)) == hipSuccess &&
	          hipMalloc(&d_dfs, W.dfs.size() * sizeof(uint32_t)) == hipSuccess && hipMalloc(&d_gconst, gconst.size() * sizeof(uint32_t)) == hipSuccess;
	ok = ok && hipMemcpy(d_chain, W.chain.data(), W.chain.size() * sizeof(ChainEntry), hipMemcpyHostToDevice) == hipSuccess &&
	     hipMemcpy(d_meta, W.meta.data(), (size_t)n * sizeof(PatchMeta), hipMemcpyHostToDevice) == hipSuccess &&
	     hipMemcpy(d_dfs, W.dfs.data(), W.dfs.size() * sizeof(uint32_t), hipMemcpyHostToDevice) == hipSuccess &&
	     hipMemcpy(d_gconst, gconst.data(), gconst.size() * sizeof(uint32_t), hipMemcpyHostToDevice) == hipSuccess;
	if (!ok) {
		if (d_chain) (void)hipFree(d_chain);
		if (d_meta) (void)hipFree(d_meta);
		if (d_dfs) (void)hipFree(d_dfs);
		if (d_gconst) (void)hipFree(d_gconst);
		return fail(std::string(who) + ": device allocation or upload of the path tables failed");
	}
	P->d_chain = d_chain; P->d_meta = d_meta; P->d_dfs = d_dfs; P->d_gconst = d_gconst;
	P->n_groups = n_groups;
	P->chain_entries = W.chain.size();
	P->plane_delta = W.flags_on ? W.delta : -1.0;
	P->graze_cos_j = P->graze_cos_i = graze_cos;
	P->ff_ready = true;
	if (env_uint("RTMI_VERBOSE", 0))
		std::fprintf(stderr, "%s: path tables of %u patches: read back %.1f ms, walk %.1f ms, group constants %.1f ms, upload %.1f ms (%zu chain entries)\n", who, n,
		             ms_read, ms_walk - ms_read, ms_consts - ms_walk, ms_since() - ms_consts, W.chain.size());
	return 0;
}

} // namespace rtmi

using namespace rtmi;

extern "C" {

int rtmi_form_factors_dev(rtmi_patches *P, const double *light_center, uint32_t samples, uint32_t light_samples,
                          uint64_t seed, uint32_t flags, uint32_t row0, uint32_t n_rows, float *d_F, uint64_t ld,
                          void *stream) {
	const char *who = "rtmi_form_factors_dev";
	if (check_patches(P, who)) return -1;
	const uint32_t n = P->n + (light_center ? 1u : 0u);
	if (n_rows == 0) return 0;
	if (!d_F) return fail(std::string(who) + ": d_F is NULL");
	if (samples == 0) return fail(std::string(who) + ": samples must be >= 1");
	if ((uint64_t)row0 + n_rows > n) return fail(std::string(who) + ": row range exceeds n");
	if (ld < n) return fail(std::string(who) + ": ld < n");
	hipStream_t st = (hipStream_t)stream;
	const rtmi_scene *s = P->scene;
	// Two kernels give the same matrix bit for bit: the path walk (this file's header) and the root traversal of round 1.
	// The first wins where the rays of a pair pass few other triangles (rooms: 37 against 10 G pair-samples/s on the 65 712-patch
	// box), the second where every ray passes many (a furnished room, a statue seen from inside: 4.2 against 9.8 and 1.5 against
	// 3.9) — there nearly every sibling box is met, each is entered on its own, and nothing is gained over walking front to back
	// from the root.  Which one serves a patch set is MEASURED at its first sizeable call: both run on the call's first rows
	// (up to 1024 of them; they are written more than once, same bits), HIP events time them, the faster stays (that one call waits for the events).
	// RTMI_FF_KERNEL=path|flat forces one.
	auto launch_flat = [&](uint32_t first, uint32_t count, float *out) -> int {
		// one lane per entry; a call with more entries than one grid holds goes out in row chunks (same stream, same bits)
		const uint32_t rows_max = (uint32_t)std::min<uint64_t>(0xffffffffull, std::max<uint64_t>(1, (0x7fffffffull * kBlock) / std::max(1u, P->n)));
		for (uint32_t done = 0; done < count;) {
			const uint32_t rows = std::min(rows_max, count - done);
			const uint64_t total = (uint64_t)rows * P->n, grid = (total + kBlock - 1) / kBlock;
			hipLaunchKernelGGL(k_form_factors_flat, dim3((uint32_t)grid), dim3(kBlock), s->stack_bytes_binary, st, s->dev, pdev(P), samples, seed, first + done,
			                   rows, out + (uint64_t)done * ld, ld);
			if (launch_check(who)) return -1;
			done += rows;
		}
		return 0;
	};
	auto launch_path = [&](uint32_t first, uint32_t count, float *out) -> int {
		uint32_t band = std::max(1u, env_uint("RTMI_FF_BAND", 64));
		while ((count + band - 1) / band > 4u * 65535u) band *= 2;
		// beam margin: the 1e-5 origin offset, the exact test's per-ray pad (<= 2^-17·absmax) and the fp32 roundings of both
		const float margin = 1.52587890625e-05f * s->dev.absmax + 1e-4f;
		const uint32_t n_bands = (count + band - 1) / band;
#if RTMI_FF_XCD_BLOCK
		const uint32_t xcd_blocks = (P->n_groups + RTMI_FF_XCD_BLOCK - 1u) / RTMI_FF_XCD_BLOCK; // blocks of groups, dealt to the 8 XCDs in turn
		const dim3 grid((xcd_blocks + 7u) / 8u * RTMI_FF_XCD_BLOCK * 8u, (n_bands + kBlock / 64 - 1) / (kBlock / 64));
#else
		const dim3 grid((P->n_groups + 7u) / 8u * 8u, (n_bands + kBlock / 64 - 1) / (kBlock / 64));
#endif
		// a lane's stack: up to pend_cap sibling subtrees waiting (more is rare: that ray starts over from the root) + the
		// depth + 1 entries the traversal inside one of them can need
		uint32_t pend_cap = std::min(s->info.max_depth + 1u, std::max(1u, env_uint("RTMI_FF_PEND", 8)));
		const uint32_t walk_bytes = RTMI_FF_WIDE_WALK ? s->stack_bytes : (s->info.max_depth + 1u) * kBlock * (uint32_t)sizeof(int);
		while (pend_cap > 1u && walk_bytes + pend_cap * kBlock * sizeof(int) > 64u * 1024u) --pend_cap; // (a workgroup's 64 KB: a very deep hierarchy leaves fewer waiting entries; rays with more start over from the root)
		// (s->stack_bytes: the deeper of the two hierarchies' paths — the rays walk the 4-wide one, the beams the binary one)
		const uint32_t lds = walk_bytes + pend_cap * kBlock * (uint32_t)sizeof(int);
		hipLaunchKernelGGL(k_form_factors, grid, dim3(kBlock), lds, st, s->dev, (const double *)P->d_verts, (const double *)P->d_normals,
		                   (const double *)P->d_area, P->n, static_cast<const ChainEntry *>(P->d_chain), static_cast<const PatchMeta *>(P->d_meta),
		                   P->d_dfs, static_cast<const uint4 *>(P->d_gconst), samples, seed, first, count, band, margin, P->graze_cos_i, P->graze_cos_j, P->plane_delta >= 0.0 ? 1u : 0u, pend_cap, P->n_groups, out, ld);
		return launch_check(who);
	};
	int kernel = 0;
	uint32_t done_rows = 0; // leading rows of this call that the kernel choice below has already written
	const char *mode = std::getenv("RTMI_FF_KERNEL");
	if (mode && std::strcmp(mode, "flat") == 0) kernel = 2;
	else if (mode && std::strcmp(mode, "path") == 0) kernel = 1;
	if (kernel != 2) {
		try {
			if (build_ff_tables(P, who)) return -1;
		} catch (const std::exception &e) {
			return fail(std::string(who) + ": building the path tables: " + e.what());
		}
	}
	if (kernel == 0) {
		std::lock_guard<std::mutex> lock(P->ff_mutex);
		kernel = P->ff_kernel;
		// rows that reach patch rows only (the light's row is zero work), enough of them for a stable time
		const uint32_t rows_p = row0 < P->n ? std::min(n_rows, P->n - row0) : 0u;
		// At least 256 rows (four bands: every wave of the path walk's workgroups has work), up to 1024: the path walk's
		// workgroups differ a lot in length (a 256-row launch is one round of them and ends with its slowest: 14-20 ms on the
		// box where 4096 rows take 72), so a short sample flatters the root traversal; 1024 rows are four rounds.
		// (not while the stream is being captured into a graph: the measurement waits for events, which a capture forbids —
		// such a call takes the path walk, or what an earlier call or RTMI_FF_KERNEL settled)
		hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
		const bool capturing = hipStreamIsCapturing(st, &capture) != hipSuccess || capture != hipStreamCaptureStatusNone;
		if (kernel == 0 && !capturing && P->n >= 1024 && rows_p >= 256) {
			const uint32_t cal = std::min(rows_p, 1024u);
			// (the root traversal has one lane per entry and workgroups of even length, and it is the slower kernel wherever the path
			// walk serves at all: slices of at least 2^21 entries — 8 192 workgroups, four to five rounds of the device; slices of
			// 2^20 made it look 1.8 times slower than it is — from both ends of the sample time it as well as the 256 rows it had
			// until round 3, which were 17 ms of this call's 93 on the 64 K box.  64 K patches: 2 slices of 32 rows; 14 676: one of
			// 143; below 8 192 patches: the 256 rows as before.)
			const uint32_t slice = std::min(256u, std::max(16u, (uint32_t)(((1ull << 21) + P->n - 1u) / P->n)));
			const uint32_t n_slices = std::max(1u, std::min(4u, 64u / slice)), cal_flat = slice * n_slices;
			hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
			bool ok = hipEventCreate(&ev[0]) == hipSuccess && hipEventCreate(&ev[1]) == hipSuccess && hipEventCreate(&ev[2]) == hipSuccess;
			float t_path = 0.f, t_flat = 0.f;
			// (an untimed run first, a quarter of the sample: after idle time the first long kernel runs at clocks that are still ramping up)
			ok = ok && launch_path(row0, cal / 4u, d_F) == 0;
			ok = ok && hipEventRecord(ev[0], st) == hipSuccess && launch_path(row0, cal, d_F) == 0 && hipEventRecord(ev[1], st) == hipSuccess;
			for (uint32_t k = 0; ok && k < n_slices; ++k) {
				const uint32_t at = n_slices > 1u ? (uint32_t)((uint64_t)(cal - slice) * k / (n_slices - 1u)) : 0u;
				ok = launch_flat(row0 + at, slice, d_F + (uint64_t)at * ld) == 0; // (the same bits the path walk just wrote there)
			}
			ok = ok && hipEventRecord(ev[2], st) == hipSuccess && hipEventSynchronize(ev[2]) == hipSuccess &&
			     hipEventElapsedTime(&t_path, ev[0], ev[1]) == hipSuccess && hipEventElapsedTime(&t_flat, ev[1], ev[2]) == hipSuccess;
			t_flat *= (float)cal / (float)cal_flat;
			done_rows = cal; // rows [row0, row0 + cal) hold the result already, whichever kernel is chosen: both write the same bits
			for (hipEvent_t e : ev)
				if (e) (void)hipEventDestroy(e);
			if (!ok) return fail(std::string(who) + ": timing the two form-factor kernels failed: " + hipGetErrorString(hipGetLastError()));
			kernel = P->ff_kernel = (t_flat * 1.1f < t_path) ? 2 : 1;
			if (env_uint("RTMI_VERBOSE", 0))
				std::fprintf(stderr, "%s: %u rows: path walk %.3f ms, root traversal %.3f ms -> %s\n", who, cal, t_path, t_flat,
				             kernel == 1 ? "path walk" : "root traversal");
		}
		if (kernel == 0) kernel = 1; // small patch sets and small first calls: the path walk, not remembered
	}
	if (done_rows < n_rows &&
	    (kernel == 2 ? launch_flat(row0 + done_rows, n_rows - done_rows, d_F + (uint64_t)done_rows * ld) : launch_path(row0 + done_rows, n_rows - done_rows, d_F + (uint64_t)done_rows * ld)) != 0)
		return -1;
	if (light_center) {
		hipLaunchKernelGGL(k_ff_light_column, dim3((n_rows + kBlock - 1) / kBlock), dim3(kBlock), s->stack_bytes_binary, st, s->dev, pdev(P),
		                   light_center[0], light_center[1], light_center[2], light_samples, seed, flags, row0, n_rows, d_F, ld);
		if (launch_check(who)) return -1;
	}
	return 0;
}

// 0 = not measured yet, 1 = path walk, 2 = root traversal (see rtmi_form_factors_dev)
int rtmi_debug_form_factor_kernel(rtmi_patches *P) {
	if (check_patches(P, "rtmi_debug_form_factor_kernel")) return -1;
	std::lock_guard<std::mutex> lock(P->ff_mutex);
	return P->ff_kernel;
}

// 1 = the path tables of this patch set carry "behind the patch's plane" flags, 0 = they do not (normals that are not the
// geometric ones, coordinates too large, RTMI_FF_PLANES=0), -1 = no tables yet (or an error)
int rtmi_debug_form_factor_planes(rtmi_patches *P) {
	if (check_patches(P, "rtmi_debug_form_factor_planes")) return -1;
	std::lock_guard<std::mutex> lock(P->ff_mutex);
	return P->ff_ready ? (P->plane_delta >= 0.0 ? 1 : 0) : -1;
}

#ifdef RTMI_FF_COUNT
// diagnostic builds: reads and clears the work counters
int rtmi_debug_ff_counts(uint64_t out[96]) {
	unsigned long long z[96] = {0};
	if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ff_counts), sizeof z) != hipSuccess) return fail("rtmi_debug_ff_counts: copy failed");
	if (hipMemcpyToSymbol(HIP_SYMBOL(g_ff_counts), z, sizeof z) != hipSuccess) return fail("rtmi_debug_ff_counts: clear failed");
	return 0;
}
#endif

} // extern "C"
