// librtmi — scene upload, closest-hit / any-hit / primary-ray kernels and their C ABI (include/rtmi.h).
// gfx950 (MI355X) only; build: see raytracer_amd/csrc/Makefile.
#include "rtmi_host.hpp"

#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

namespace rtmi {

thread_local std::string g_err;

int fail(const std::string &msg) {
	g_err = msg;
	return -1;
}

// ------------------------------------------------------------------------------------------- kernels
// HitQ -> the record of include/rtmi.h (original id and object come from the triangle the lane remembered)
__device__ __forceinline__ rtmi_hit hit_record(const SceneDev &S, const HitQ &h) {
	rtmi_hit o;
	o.t = h.t; o.u = h.u; o.v = h.v; o.tri = RTMI_MISS; o.object = RTMI_MISS;
	if (h.pos != RTMI_MISS) {
		const uint2 ids = *reinterpret_cast<const uint2 *>(&S.tris[h.pos].orig);
		o.tri = ids.x; o.object = ids.y;
	}
	return o;
}

// Rays from a buffer: the launch's waves work through rays [0, n) as ONE queue (a counter in device memory)
// (trace_queue); callers that care about coherence sort the buffer first.
#ifndef RTMI_AO_WAVES
#define RTMI_AO_WAVES 4
#endif
template <bool ANY>
RTMI_TRAVERSAL_KERNEL void k_trace_rays(SceneDev S, const rtmi_ray *__restrict__ rays,
                                                       uint64_t n, const uint32_t *__restrict__ d_n, uint32_t mul,
                                                       int *__restrict__ counter, Tune tune, rtmi_hit *__restrict__ out,
                                                       uint8_t *__restrict__ occluded) {
	extern __shared__ int stack[];
	const uint64_t base = 0;
	// d_n: the number of rays is known only on the device (a compacted wavefront): min(n, *d_n · mul)
	const uint32_t items = d_n ? (uint32_t)min((unsigned long long)n, (unsigned long long)*d_n * mul) : (uint32_t)n;
	auto fetch = [&](uint32_t idx, RayD &R) {
		const rtmi_ray q = rays[base + idx];
		if (q.reserved != 0.0) { // an inactive slot of a wavefront (e.g. the shadow ray of a pixel that hit nothing)
			if (ANY) occluded[base + idx] = 0;
			else { rtmi_hit o; o.t = 0.0; o.u = o.v = 0.f; o.tri = o.object = RTMI_MISS; out[base + idx] = o; }
			return false;
		}
		R.ox = q.o[0]; R.oy = q.o[1]; R.oz = q.o[2];
		R.dx = q.d[0]; R.dy = q.d[1]; R.dz = q.d[2];
		double tmax = q.tmax;
		if (!(tmax > 0.0) || tmax > RTMI_INFINITY) tmax = RTMI_INFINITY;
		R.tmax = tmax;
		return true;
	};
	auto emit = [&](uint32_t idx, const HitQ &h, RayD &) {
		if (ANY) occluded[base + idx] = h.pos != RTMI_MISS ? 1 : 0;
		else out[base + idx] = hit_record(S, h);
		return false;
	};
	trace_queue<ANY, false, false>(S, stack + threadIdx.x, counter, items, tune, 0.0, 0.0, 0.0, fetch, emit);
}

// Shader::findOcclusionNew (shader.cpp:140-232) in ONE launch: the ambient-occlusion ray of item k·spp + s is built in
// the refill of the any-hit queue — hit point (object.cpp:23), hemisphere frame, square point of the stream
// (seed, k, s), exactly the operations of k_ao_rays (rtmi_wavefront.hip) — and never exists in memory: 64 B written and
// 64 B read per ray less than rtmi_ao_rays_dev + rtmi_any_hit_dev, same bytes out.
RTMI_TRAVERSAL_KERNEL_W(RTMI_AO_WAVES) void k_ao_occlusion(SceneDev S, const double *__restrict__ v1_orig, const double *__restrict__ normals,
                                             const rtmi_ray *__restrict__ rays, const rtmi_hit *__restrict__ hits, uint32_t n_items,
                                             uint32_t spp, double radius, uint64_t seed, int *__restrict__ counter, Tune tune,
                                             uint8_t *__restrict__ occluded) {
	extern __shared__ int stack[];
	auto fetch = [&](uint32_t idx, RayD &R) {
		const uint32_t k = idx / spp, smp = idx % spp;
		const rtmi_hit h = hits[k];
		if (h.tri == RTMI_MISS) {
			occluded[idx] = 0;
			return false;
		}
		const D3 nrm = d3(normals[3 * (size_t)h.tri], normals[3 * (size_t)h.tri + 1], normals[3 * (size_t)h.tri + 2]);
		const D3 P = hit_point(rays[k], v1_orig + 3 * (size_t)h.tri, nrm);
		D3 org, dir;
		ao_ray(P, nrm, radius, wf_mix64(wf_mix64(seed ^ (uint64_t)k) ^ (uint64_t)smp), org, dir);
		R.ox = org.x; R.oy = org.y; R.oz = org.z;
		R.dx = dir.x; R.dy = dir.y; R.dz = dir.z;
		R.tmax = radius; // occluded ⇔ Cast ∧ distance < R (:226)
		return true;
	};
	auto emit = [&](uint32_t idx, const HitQ &h, RayD &) {
		occluded[idx] = h.pos != RTMI_MISS ? 1 : 0;
		return false;
	};
	trace_queue<true, false, false>(S, stack + threadIdx.x, counter, n_items, tune, 0.0, 0.0, 0.0, fetch, emit);
}

// Shader::generateNormalShadows (shader.cpp:105-138) in one launch: the shadow ray of hit k (origin P + 0.01·normal, direction
// Unit(C − P), :116-127) is built in the refill of the any-hit queue; occluded[k] = Cast's verdict (:131).  The operations are
// those of k_shadow_rays (rtmi_wavefront.hip).
RTMI_TRAVERSAL_KERNEL_W(RTMI_AO_WAVES) void k_shadow_occlusion(SceneDev S, const double *__restrict__ v1_orig, const double *__restrict__ normals,
                                                 const rtmi_ray *__restrict__ rays, const rtmi_hit *__restrict__ hits, uint32_t n_items,
                                                 double cx, double cy, double cz, int *__restrict__ counter, Tune tune,
                                                 uint8_t *__restrict__ occluded) {
	extern __shared__ int stack[];
	auto fetch = [&](uint32_t k, RayD &R) {
		const rtmi_hit h = hits[k];
		if (h.tri == RTMI_MISS) {
			occluded[k] = 0;
			return false;
		}
		const D3 nrm = d3(normals[3 * (size_t)h.tri], normals[3 * (size_t)h.tri + 1], normals[3 * (size_t)h.tri + 2]);
		const D3 P = hit_point(rays[k], v1_orig + 3 * (size_t)h.tri, nrm);
		const D3 dir = unit(d3(cx, cy, cz) - P);
		const D3 org = P + (0.01 * nrm);
		R.ox = org.x; R.oy = org.y; R.oz = org.z;
		R.dx = dir.x; R.dy = dir.y; R.dz = dir.z;
		R.tmax = RTMI_INFINITY;
		return true;
	};
	auto emit = [&](uint32_t k, const HitQ &h, RayD &) {
		occluded[k] = h.pos != RTMI_MISS ? 1 : 0;
		return false;
	};
	trace_queue<true, false, false>(S, stack + threadIdx.x, counter, n_items, tune, 0.0, 0.0, 0.0, fetch, emit);
}

// Shader::generateSoftShadows (shader.cpp:41-103) in one launch, like k_ao_occlusion: the ray of item k·n_points + i·divY + j
// toward grid point (i, j) through the light centre (:60-71, :90-91) is built in the refill of a closest-hit queue, and
// only its verdict is stored: 1 iff it hits a triangle of ANOTHER object than the shaded hit's (:93-100).  The operations
// are those of k_soft_shadow_rays + k_soft_shadow_factor (rtmi_wavefront.hip).
RTMI_TRAVERSAL_KERNEL_W(RTMI_AO_WAVES) void k_soft_shadow(SceneDev S, const double *__restrict__ v1_orig, const double *__restrict__ normals,
                                            const rtmi_ray *__restrict__ rays, const rtmi_hit *__restrict__ hits, uint32_t n_items,
                                            uint32_t n_points, uint32_t div_x, uint32_t div_y, double cx, double cy, double cz,
                                            int *__restrict__ counter, Tune tune, uint8_t *__restrict__ contrib) {
	extern __shared__ int stack[];
	auto fetch = [&](uint32_t idx, RayD &R) {
		const uint32_t k = idx / n_points, smp = idx % n_points;
		const rtmi_hit h = hits[k];
		if (h.tri == RTMI_MISS) {
			contrib[idx] = 0;
			return false;
		}
		const D3 nrm = d3(normals[3 * (size_t)h.tri], normals[3 * (size_t)h.tri + 1], normals[3 * (size_t)h.tri + 2]);
		const D3 P = hit_point(rays[k], v1_orig + 3 * (size_t)h.tri, nrm);
		const uint32_t i = smp / div_y, j = smp % div_y;
		const D3 target = d3(cx - 1 + ((j * 2 + 1) / (double)div_y), cy - 1 + ((i * 2 + 1) / (double)div_x), cz); // :69
		const D3 dir = unit(target - P);
		const D3 org = P + (0.05 * dir);
		R.ox = org.x; R.oy = org.y; R.oz = org.z;
		R.dx = dir.x; R.dy = dir.y; R.dz = dir.z;
		R.tmax = RTMI_INFINITY;
		return true;
	};
	auto emit = [&](uint32_t idx, const HitQ &h, RayD &) {
		contrib[idx] = (h.pos != RTMI_MISS && S.tris[h.pos].object != hits[idx / n_points].object) ? 1 : 0;
		return false;
	};
	trace_queue<false, false, false>(S, stack + threadIdx.x, counter, n_items, tune, 0.0, 0.0, 0.0, fetch, emit);
}

// factor = 1 − contributors / n_points, 1 when there is none (shader.cpp:100-102)
__global__ void k_soft_shadow_reduce(const uint8_t *__restrict__ contrib, uint64_t n, uint32_t n_points, double *__restrict__ out) {
	const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= n) return;
	int c = 0;
	for (uint32_t s = 0; s < n_points; ++s) c += contrib[k * n_points + s];
	out[k] = c > 0 ? (1.0 - ((double)c) / ((double)n_points)) : 1.0;
}

// Render2/rasterizer.cpp:230 — dir(i,j) = Unit(O + (j+sx)·dR − (i+sy)·dU), origin = eye.
__device__ __forceinline__ D3 primary_dir(const rtmi_camera &c, uint32_t i, uint32_t j, double sx, double sy) {
	const D3 O = d3(c.O[0], c.O[1], c.O[2]), dR = d3(c.dR[0], c.dR[1], c.dR[2]), dU = d3(c.dU[0], c.dU[1], c.dU[2]);
	return unit((O + (j + sx) * dR) - (i + sy) * dU);
}

// Division of a work-item index (< 2^31) by a launch constant: q = mulhi(n, m) >> k with m = ceil(2^(32+k) / d), k = floor(log2 d)
// (exact for n < 2^31: m·d = 2^(32+k) + e with e < d <= 2^(k+1), so the error n·e / (d·2^(32+k)) stays below 1/d); powers of two
// (m = 0) shift.  The compiler's division by a run-time value is ≈ 28 instructions; an item -> pixel mapping has five of them, and
// it runs when a ray is fetched and again when its record is written — at the lane occupancy of a refill.
struct FastDiv { uint32_t d, m, k; };
static FastDiv make_fastdiv(uint32_t d) {
	FastDiv f = {d ? d : 1u, 0u, 0u};
	while ((2u << f.k) <= f.d && f.k < 31u) ++f.k; // k = floor(log2 d)
	if ((f.d & (f.d - 1u)) != 0u) f.m = (uint32_t)((((unsigned long long)1 << (32 + f.k)) + f.d - 1u) / f.d);
	return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv &f) { return f.m ? (__umulhi(n, f.m) >> f.k) : (n >> f.k); }

struct PrimaryGrid {
	uint32_t tile_w, tile_h, tiles_x, tile_first, tile_stride, n_tiles;
	uint32_t blk_w, blk_h;            // pixels one workgroup owns (multiples of 8)
	uint32_t blocks_x, blocks_per_tile;
	FastDiv d_per_block, d_blocks_per_tile, d_tiles_x, d_blocks_x, d_sq_x; // blk_w·blk_h, blocks_per_tile, tiles_x, blocks_x, blk_w / 8
};

// item k of a workgroup's blk_w × blk_h pixel block: consecutive runs of 64 items are 8×8 pixel squares, so
// the rays a wave claims together are neighbours and walk the same nodes.
__device__ __forceinline__ bool primary_pixel(const PrimaryGrid &g, const rtmi_camera &cam, uint32_t item, uint32_t &i, uint32_t &j) {
	const uint32_t block = fdiv(item, g.d_per_block), k = item - block * g.d_per_block.d;
	const uint32_t local_tile = fdiv(block, g.d_blocks_per_tile), b = block - local_tile * g.blocks_per_tile;
	const uint32_t tile = g.tile_first + local_tile * g.tile_stride;
	const uint32_t ty = fdiv(tile, g.d_tiles_x), tx = tile - ty * g.tiles_x;
	const uint32_t sq = k >> 6, in = k & 63u;
	const uint32_t by = fdiv(b, g.d_blocks_x), bx = b - by * g.blocks_x;
	const uint32_t sy = fdiv(sq, g.d_sq_x), sx = sq - sy * g.d_sq_x.d;
	const uint32_t lx = bx * g.blk_w + sx * 8 + (in & 7u);
	const uint32_t ly = by * g.blk_h + sy * 8 + (in >> 3);
	j = tx * g.tile_w + lx;
	i = ty * g.tile_h + ly;
	return tile < g.n_tiles && lx < g.tile_w && ly < g.tile_h && j < cam.x_res && i < cam.y_res;
}

// Primary rays generated in registers, 0 input bytes per ray.
RTMI_TRAVERSAL_KERNEL void k_primary_closest(SceneDev S, int *__restrict__ counter, uint32_t n_items, rtmi_camera cam,
                                                            double sx, double sy, PrimaryGrid g, Tune tune,
                                                            rtmi_hit *__restrict__ out) {
	extern __shared__ int stack[];
	auto fetch = [&](uint32_t idx, RayD &R) {
		uint32_t i, j;
		if (!primary_pixel(g, cam, idx, i, j)) return false;
		const D3 d = primary_dir(cam, i, j, sx, sy);
		R.dx = d.x; R.dy = d.y; R.dz = d.z;
		R.tmax = RTMI_INFINITY;
		return true;
	};
	auto emit = [&](uint32_t idx, const HitQ &h, RayD &) {
		uint32_t i, j;
		primary_pixel(g, cam, idx, i, j);
		out[(uint64_t)i * cam.x_res + j] = hit_record(S, h);
		return false;
	};
	trace_queue<false, false, true>(S, stack + threadIdx.x, counter, n_items, tune, cam.eye[0], cam.eye[1], cam.eye[2],
	                                fetch, emit);
}

// Several samples per pixel, one queue: item -> (sample, item of that sample's tiles).  The sample table is a kernel argument
// (constant memory), read per lane when a ray is fetched and when its record is written.
struct SampleSet {
	rtmi_primary_sample s[RTMI_MAX_SAMPLES_PER_LAUNCH];
	uint32_t n, items_per_sample;
	FastDiv d_items_per_sample;
};

RTMI_TRAVERSAL_KERNEL void k_primary_closest_samples(SceneDev S, int *__restrict__ counter, uint32_t n_items, rtmi_camera cam,
                                                                    PrimaryGrid g, Tune tune, SampleSet set) {
	extern __shared__ int stack[];
	__shared__ rtmi_primary_sample table[RTMI_MAX_SAMPLES_PER_LAUNCH];
	for (uint32_t k = 0; k < RTMI_MAX_SAMPLES_PER_LAUNCH; ++k) // uniform index into the argument, one lane stores
		if (threadIdx.x == k && k < set.n) table[k] = set.s[k];
	__syncthreads();
	auto pixel = [&](uint32_t idx, uint32_t &smp, uint32_t &i, uint32_t &j) {
		smp = fdiv(idx, set.d_items_per_sample);
		const uint32_t k = idx - smp * set.items_per_sample;
		PrimaryGrid gs = g;
		gs.tile_first = table[smp].tile_first;
		return primary_pixel(gs, cam, k, i, j);
	};
	auto fetch = [&](uint32_t idx, RayD &R) {
		uint32_t smp, i, j;
		if (!pixel(idx, smp, i, j)) return false;
		const D3 d = primary_dir(cam, i, j, table[smp].sx, table[smp].sy);
		R.dx = d.x; R.dy = d.y; R.dz = d.z;
		R.tmax = RTMI_INFINITY;
		return true;
	};
	auto emit = [&](uint32_t idx, const HitQ &h, RayD &) {
		uint32_t smp, i, j;
		pixel(idx, smp, i, j);
		table[smp].d_out[(uint64_t)i * cam.x_res + j] = hit_record(S, h);
		return false;
	};
	trace_queue<false, false, true>(S, stack + threadIdx.x, counter, n_items, tune, cam.eye[0], cam.eye[1], cam.eye[2],
	                                fetch, emit);
}

// Same rays as k_primary_closest (all tiles); only counts the traversal work: counts[0] += node visits
// (one visit = two child boxes), counts[1] += exact triangle tests, counts[2..3] += the lane slots the wave spent on
// them (64 per wave-iteration: visits / slots = lane occupancy of that loop).  Diagnostic, not on the hot path.
RTMI_TRAVERSAL_KERNEL void k_primary_work(SceneDev S, int *__restrict__ counter, uint32_t n_items, rtmi_camera cam, double sx,
                                                         double sy, PrimaryGrid g, Tune tune, unsigned long long *__restrict__ counts) {
	extern __shared__ int stack[];
	const uint32_t lane = threadIdx.x & 63u;
	Work w = {0u, 0u, 0u, 0u, 0u, 0u};
	auto fetch = [&](uint32_t idx, RayD &R) {
		uint32_t i, j;
		if (!primary_pixel(g, cam, idx, i, j)) return false;
		const D3 d = primary_dir(cam, i, j, sx, sy);
		R.dx = d.x; R.dy = d.y; R.dz = d.z;
		R.tmax = RTMI_INFINITY;
		return true;
	};
	auto emit = [&](uint32_t, const HitQ &, RayD &) { return false; };
	trace_queue<false, true, true>(S, stack + threadIdx.x, counter, n_items, tune, cam.eye[0], cam.eye[1], cam.eye[2],
	                               fetch, emit, &w);
	unsigned long long v[6] = {w.nodes, w.tests, w.node_slots, w.test_slots, w.wait_leaf, w.wait_idle};
	for (int k = 0; k < 6; ++k) {
		for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_xor(v[k], off, 64);
		if (lane == 0) atomicAdd(&counts[k], v[k]);
	}
}

__global__ void k_primary_rays(rtmi_camera cam, double sx, double sy, rtmi_ray *__restrict__ rays) {
	const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const uint64_t n = (uint64_t)cam.x_res * cam.y_res;
	if (k >= n) return;
	const uint32_t i = (uint32_t)(k / cam.x_res), j = (uint32_t)(k % cam.x_res);
	const D3 d = primary_dir(cam, i, j, sx, sy);
	rtmi_ray R;
	R.o[0] = cam.eye[0]; R.o[1] = cam.eye[1]; R.o[2] = cam.eye[2];
	R.d[0] = d.x; R.d[1] = d.y; R.d[2] = d.z;
	R.tmax = RTMI_INFINITY;
	R.reserved = 0.0;
	rays[k] = R;
}

// Render2/object.cpp:29-60 on n independent (ray, triangle) pairs: the known-answer entry point.
__global__ void k_triangle_intersection(uint64_t n, const double *__restrict__ o, const double *__restrict__ d,
                                        const double *__restrict__ tri, double *__restrict__ t, double *__restrict__ u,
                                        double *__restrict__ v) {
	const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= n) return;
	GTri T;
	for (int a = 0; a < 3; ++a) {
		T.v0[a] = tri[9 * k + a];
		T.e1[a] = tri[9 * k + 3 + a] - tri[9 * k + a];
		T.e2[a] = tri[9 * k + 6 + a] - tri[9 * k + a];
	}
	double uu = u[k], vv = v[k]; // the reference leaves u, v untouched on the early exits
	t[k] = mt_exact(o[3 * k], o[3 * k + 1], o[3 * k + 2], d[3 * k], d[3 * k + 1], d[3 * k + 2], T, uu, vv);
	u[k] = uu;
	v[k] = vv;
}

// ------------------------------------------------------------------------------------------- scene
unsigned env_uint(const char *name, unsigned dflt) {
	const char *s = std::getenv(name);
	if (!s || !*s) return dflt;
	long v = std::strtol(s, nullptr, 10);
	return v > 0 ? (unsigned)v : dflt;
}

#ifdef RTMI_NODE_STRIDE
static const size_t kNodeStride = RTMI_NODE_STRIDE;
#else
static const size_t kNodeStride = sizeof(WNode);
#endif
Tune refill_threshold() { // the tuning environment is read once
	static const Tune t = [] {
		Tune x;
		x.thresh = (int)std::min(64u, env_uint("RTMI_REFILL", 16));
		x.vote_node = (int)env_uint("RTMI_VOTE_NODE", 1);
		x.vote_leaf = (int)env_uint("RTMI_VOTE_LEAF", 3);
		x.chunk = kQueueChunk;
		x.tail_chunk = kTailChunk;
		x.max_chunk = std::max(64u, env_uint("RTMI_MAX_CHUNK", 512));
		return x;
	}();
	return t;
}

// Claim sizes a launch STARTS with (trace_queue then sizes its claims from the rate at which its workgroups use items, between
// 64 and max_chunk, and from what is left of the queue): kQueueChunk items per atomic, half of that when the launch is so small
// that this would hand every resident wave its whole share in one or two claims (a launch cannot end before its slowest wave
// has; smaller first claims leave something to balance with).  tail_chunk is the floor of the last claims.
Tune launch_tune(Tune t, uint32_t grid, uint64_t n_items) {
	static const unsigned fixed = env_uint("RTMI_CHUNK", 0);
	if (fixed) { t.chunk = fixed; t.tail_chunk = std::max(16u, fixed / 2u); t.max_chunk = fixed; return t; }
	const uint64_t per_wave = n_items / std::max<uint64_t>(1, (uint64_t)grid * (kBlock / 64));
	t.chunk = per_wave < kQueueChunk ? kQueueChunk / 2u : kQueueChunk;
	t.tail_chunk = 32u;
	return t;
}

static rtmi_scene *scene_create(const double *xyz, const uint32_t *obj, uint64_t n) {
	if (n > 0 && !xyz) { fail("rtmi_scene_create: tri_xyz is NULL"); return nullptr; }
	if (n >= kMaxFirst) { fail("rtmi_scene_create: more than 2^27-1 triangles"); return nullptr; }
	for (uint64_t i = 0; i < 9 * n; ++i)
		if (!std::isfinite(xyz[i])) { fail("rtmi_scene_create: non-finite vertex coordinate"); return nullptr; }
	auto t0 = std::chrono::steady_clock::now();
	// RTMI_VERBOSE=1: seconds per phase of the build on stderr
	const bool verbose = env_uint("RTMI_VERBOSE", 0) != 0;
	auto t_phase = t0;
	auto phase = [&](const char *what) {
		const auto now = std::chrono::steady_clock::now();
		if (verbose) std::fprintf(stderr, "rtmi_scene_create: %-28s %.3f s\n", what, std::chrono::duration<double>(now - t_phase).count());
		t_phase = now;
	};
	const unsigned leaf = 1; // the 4-wide nodes address a leaf child as ONE triangle (rtmi_device.hpp, WNode)
	const unsigned threads = std::min(env_uint("RTMI_BUILD_THREADS", std::max(1u, std::thread::hardware_concurrency())), 16u);
	Bvh2 bvh = build_bvh2(xyz, n, leaf, kStackDepth - 2, threads);
	if (n > 0 && bvh.depth > (uint32_t)kStackDepth - 1) { fail("rtmi_scene_create: BVH deeper than the traversal stack"); return nullptr; }
	phase("binary hierarchy (SAH)");

	rtmi_scene *s = new rtmi_scene();
	s->n_tris = n;
	s->n_objects = 1;
	if (obj) for (uint64_t i = 0; i < n; ++i) s->n_objects = std::max(s->n_objects, obj[i] + 1u);
	s->info.n_tris = n;
	s->info.n_leaves = bvh.n_leaves;
	s->info.max_depth = bvh.depth;
	s->info.max_leaf_size = bvh.max_leaf;
	std::memcpy(s->info.scene_lo, bvh.lo, sizeof bvh.lo);
	std::memcpy(s->info.scene_hi, bvh.hi, sizeof bvh.hi);
	if (hipGetDevice(&s->device) != hipSuccess) { delete s; fail("hipGetDevice failed"); return nullptr; }
	s->dev.nodes = nullptr; s->dev.wnodes = nullptr; s->dev.tris = nullptr; s->dev.root_ref = kSentinel; s->dev.absmax = 0.f;
	for (int a = 0; a < 3; ++a) { s->dev.qlo[a] = 0.f; s->dev.qstep[a] = 0.f; }
	s->d_v1_orig = nullptr;
	s->xyz_hash = hash_doubles(xyz, 9 * n);
	s->d_queue = nullptr;
	s->h_error = nullptr;
	s->dev.n_top = 0;
	s->dev.guard = env_uint("RTMI_GUARD", kGuard);
	s->dev.error = nullptr;
	{
		hipDeviceProp_t prop;
		s->n_cus = (hipGetDeviceProperties(&prop, s->device) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
	}
	if (hipMalloc(&s->d_queue, kQueueSlots * sizeof(int)) != hipSuccess) { delete s; fail("rtmi_scene_create: hipMalloc failed"); return nullptr; }
	// the error word lives in host memory the device can write: the host reads it without a copy
	if (hipHostMalloc((void **)&s->h_error, sizeof(int), hipHostMallocMapped) != hipSuccess ||
	    hipHostGetDevicePointer((void **)&s->dev.error, s->h_error, 0) != hipSuccess) {
		if (s->h_error) (void)hipHostFree(s->h_error);
		(void)hipFree(s->d_queue);
		delete s;
		fail("rtmi_scene_create: hipHostMalloc failed");
		return nullptr;
	}
	*s->h_error = 0;
	// binary hierarchy (display pass): one push per inner node on a root-to-leaf path at most; the 4-wide one may need more
	s->stack_bytes = s->stack_bytes_binary = (bvh.depth + 1u) * kBlock * (uint32_t)sizeof(int);
	if (n == 0) return s;
	phase("hash, device set-up");

	// ---- the 4-wide hierarchy of the ray queries: collapse, lay out, quantise (bvh_build.cpp: build_wide) ----------
	Wide4 wide;
	{
		std::string err;
		if (build_wide(bvh, wide, err, threads)) { (void)hipHostFree(s->h_error); (void)hipFree(s->d_queue); delete s; fail("rtmi_scene_create: " + err); return nullptr; }
	}
	phase("4-wide hierarchy");
	static_assert(sizeof(WideNode) == sizeof(WNode), "host and device layout of the 4-wide node");
	const std::vector<uint32_t> &leaf_pos = wide.leaf_pos;
	const std::vector<WideNode> &wnodes = wide.nodes;
	const uint32_t wide_need = wide.stack_need;
	if ((wide_need + 1u) * kBlock * sizeof(int) > 64u * 1024u) { (void)hipHostFree(s->h_error); (void)hipFree(s->d_queue); delete s; fail("rtmi_scene_create: BVH deeper than the traversal stack"); return nullptr; }
	s->stack_bytes = (std::max(bvh.depth, wide_need) + 1u) * kBlock * (uint32_t)sizeof(int);
	if (verbose)
		std::fprintf(stderr, "rtmi_scene_create: %zu binary nodes of depth %u, %zu 4-wide nodes needing %u stack entries, %u B of LDS per workgroup\n",
		             bvh.nodes.size(), bvh.depth, wnodes.size(), wide_need, s->stack_bytes);

	// leaf-ordered triangles (position = leaf_pos of their leaf)
	std::vector<GTri> tris(n);
	for (size_t i = 0; i < bvh.nodes.size(); ++i) {
		const BuildNode &bn = bvh.nodes[i];
		if (bn.left >= 0) continue;
		const uint32_t id = bvh.order[bn.first];
		const double *p = xyz + 9 * (uint64_t)id;
		GTri &T = tris[leaf_pos[i]];
		for (int a = 0; a < 3; ++a) {
			T.v0[a] = p[a];
			T.e1[a] = p[3 + a] - p[a]; // Render2/object.cpp:36
			T.e2[a] = p[6 + a] - p[a]; // Render2/object.cpp:37
		}
		T.orig = id;
		T.object = obj ? obj[id] : 0u;
	}
	phase("leaf-ordered triangles");
	// ---- the binary nodes of the display pass: depth-first order (a node's near subtree follows it in memory)
	auto ref_of_leaf = [&](int32_t build_id) { return (int32_t) ~(leaf_pos[build_id] << 3); };
	// the 16-bit grid over the scene bounds: plane = qlo + q·qstep, evaluated by the kernels in fp32.  qstep is
	// rounded up until qlo + 65535·qstep covers the scene; lo planes are rounded down and hi planes up against
	// the very values the kernels use, so a quantised box always contains the fp32 box.
	float qlo[3], qstep[3];
	for (int a = 0; a < 3; ++a) {
		qlo[a] = bvh.lo[a];
		const double ext = (double)bvh.hi[a] - (double)bvh.lo[a];
		float st = (float)(ext / 65535.0);
		while (ext > 0.0 && (double)qlo[a] + 65535.0 * (double)st < (double)bvh.hi[a]) st = std::nextafterf(st, INFINITY);
		qstep[a] = st;
	}
	auto q_lo = [&](float v, int a) -> uint32_t {
		if (!(qstep[a] > 0.f)) return 0u;
		double q = std::floor(((double)v - (double)qlo[a]) / (double)qstep[a]);
		q = std::min(65535.0, std::max(0.0, q));
		while (q > 0.0 && (double)qlo[a] + q * (double)qstep[a] > (double)v) q -= 1.0;
		return (uint32_t)q;
	};
	auto q_hi = [&](float v, int a) -> uint32_t {
		if (!(qstep[a] > 0.f)) return 0u;
		double q = std::ceil(((double)v - (double)qlo[a]) / (double)qstep[a]);
		q = std::min(65535.0, std::max(0.0, q));
		while (q < 65535.0 && (double)qlo[a] + q * (double)qstep[a] < (double)v) q += 1.0;
		return (uint32_t)q;
	};
	auto plane_word = [&](const BuildNode &c, int a) { return q_lo(c.lo[a], a) | (q_hi(c.hi[a], a) << 16); };
	std::vector<QNode> nodes;
	const BuildNode &root = bvh.nodes[0];
	int32_t root_ref;
	if (root.left < 0) {
		root_ref = ref_of_leaf(0);
	} else {
		nodes.reserve(bvh.nodes.size() / 2 + 1);
		struct Item { uint32_t build_id; int32_t out_id; };
		std::vector<Item> todo;
		nodes.emplace_back();
		todo.push_back({0u, 0});
		while (!todo.empty()) {
			Item it = todo.back();
			todo.pop_back();
#if RTMI_FF_WIDE_WALK
			if (s->wide_of_node.size() <= (size_t)it.out_id) s->wide_of_node.resize((size_t)it.out_id + 1, 0xffffffffu);
			s->wide_of_node[it.out_id] = wide.cover[it.build_id];
#endif
			const BuildNode &b = bvh.nodes[it.build_id];
			const BuildNode &c0 = bvh.nodes[b.left], &c1 = bvh.nodes[b.right];
			int32_t r0, r1;
			// allocate right first so that the left child is processed (and laid out) first
			if (c1.left < 0) r1 = ref_of_leaf(b.right);
			else { r1 = (int32_t)nodes.size(); nodes.emplace_back(); todo.push_back({(uint32_t)b.right, r1}); }
			if (c0.left < 0) r0 = ref_of_leaf(b.left);
			else { r0 = (int32_t)nodes.size(); nodes.emplace_back(); todo.push_back({(uint32_t)b.left, r0}); }
			QNode g;
			g.a = make_uint4(plane_word(c0, 0), plane_word(c0, 1), plane_word(c0, 2), plane_word(c1, 0));
			g.b = make_uint4(plane_word(c1, 1), plane_word(c1, 2), (uint32_t)r0, (uint32_t)r1);
			nodes[it.out_id] = g;
		}
		root_ref = 0;
	}
	phase("binary display nodes");
	float absmax = 0.f;
	for (int a = 0; a < 3; ++a) absmax = std::max(absmax, std::max(std::fabs(bvh.lo[a]), std::fabs(bvh.hi[a])));

	std::vector<double> v1(3 * n);
	for (uint64_t k = 0; k < n; ++k)
		for (int a = 0; a < 3; ++a) v1[3 * k + a] = xyz[9 * k + a];
	QNode *d_nodes = nullptr;
	WNode *d_wnodes = nullptr;
	GTri *d_tris = nullptr;
	const size_t nb = std::max<size_t>(1, nodes.size()) * sizeof(QNode), wb = std::max<size_t>(1, wnodes.size()) * kNodeStride, tb = tris.size() * sizeof(GTri);
	if (hipMalloc(&d_nodes, nb) != hipSuccess || hipMalloc(&d_wnodes, wb) != hipSuccess || hipMalloc(&d_tris, tb) != hipSuccess ||
	    hipMalloc(&s->d_v1_orig, v1.size() * sizeof(double)) != hipSuccess) {
		if (d_nodes) (void)hipFree(d_nodes);
		if (d_wnodes) (void)hipFree(d_wnodes);
		if (d_tris) (void)hipFree(d_tris);
		if (s->d_v1_orig) (void)hipFree(s->d_v1_orig);
		(void)hipFree(s->d_queue);
		(void)hipHostFree(s->h_error);
		delete s;
		fail("rtmi_scene_create: hipMalloc failed");
		return nullptr;
	}
	bool ok = true;
	if (!nodes.empty()) ok = ok && hipMemcpy(d_nodes, nodes.data(), nodes.size() * sizeof(QNode), hipMemcpyHostToDevice) == hipSuccess;
	if (!wnodes.empty()) {
		// (one plain copy: hipMemcpy2D moved these 48-byte rows one by one — 4.6 s of a 5.3 s scene build at 1 M triangles)
		if (kNodeStride == sizeof(WNode)) {
			ok = ok && hipMemcpy(d_wnodes, wnodes.data(), wnodes.size() * sizeof(WNode), hipMemcpyHostToDevice) == hipSuccess;
		} else { // the padded layout of the RTMI_NODE_STRIDE experiment
			std::vector<unsigned char> padded(wnodes.size() * (size_t)kNodeStride, 0);
			for (size_t i = 0; i < wnodes.size(); ++i) std::memcpy(padded.data() + i * (size_t)kNodeStride, &wnodes[i], sizeof(WNode));
			ok = ok && hipMemcpy(d_wnodes, padded.data(), padded.size(), hipMemcpyHostToDevice) == hipSuccess;
		}
	}
	ok = ok && hipMemcpy(d_tris, tris.data(), tb, hipMemcpyHostToDevice) == hipSuccess;
	ok = ok && hipMemcpy(s->d_v1_orig, v1.data(), v1.size() * sizeof(double), hipMemcpyHostToDevice) == hipSuccess;
	if (!ok) {
		(void)hipFree(d_nodes); (void)hipFree(d_wnodes); (void)hipFree(d_tris); (void)hipFree(s->d_v1_orig); (void)hipFree(s->d_queue); (void)hipHostFree(s->h_error);
		delete s;
		fail("rtmi_scene_create: upload failed");
		return nullptr;
	}
	phase("device allocation and upload");
	s->dev.n_top = wide.n_top;
	s->dev.nodes = d_nodes; s->dev.wnodes = d_wnodes; s->dev.tris = d_tris; s->dev.root_ref = root_ref; s->dev.absmax = absmax;
	for (int a = 0; a < 3; ++a) { s->dev.qlo[a] = qlo[a]; s->dev.qstep[a] = qstep[a]; }
	s->info.n_nodes = nodes.size();
	s->info.device_bytes = nb + wb + tb + v1.size() * sizeof(double);
	s->info.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	return s;
}

int check_scene(const rtmi_scene *s, const char *who) {
	if (!s) return fail(std::string(who) + ": scene is NULL");
	int dev = -1;
	if (hipGetDevice(&dev) != hipSuccess) return fail(std::string(who) + ": hipGetDevice failed");
	if (dev != s->device) return fail(std::string(who) + ": the scene lives on another device than the current one");
	// sticky, like a device fault: a launch whose waves gave up left records unwritten
	if (s->h_error && *(volatile int *)s->h_error != 0)
		return fail(std::string(who) + ": an earlier launch on this scene gave up (a wave exceeded its iteration budget, RTMI_GUARD): its results are incomplete");
	return 0;
}

int launch_check(const char *who) {
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) return fail(std::string(who) + ": launch failed: " + hipGetErrorString(e));
	return 0;
}

// The queue counter of a launch: one per STREAM (launches on a stream are ordered, so they share it; it is zeroed on the
// stream just before the kernel).  Launches on different streams never share a counter, however many are in flight.
// A counter is never taken away from a stream handle while the scene lives (nothing says when a stream's last launch on
// it has drained, or that the handle is dead): a host that keeps making new streams gets further blocks of counters —
// 4 bytes per distinct handle value; the allocator reuses the values of destroyed streams, which land on their old slot.
// The first launch on the 257th, 513th, ... distinct stream therefore allocates (hipMalloc: not inside a graph capture).
constexpr uint32_t kQueueSlotsMax = 1u << 22;
int *queue_slot(rtmi_scene *s, hipStream_t st, const char *who) {
	int *slot = nullptr;
	{
		std::lock_guard<std::mutex> lock(s->queue_mutex);
		uint32_t idx;
		auto it = s->stream_slot.find((void *)st);
		if (it != s->stream_slot.end()) idx = it->second;
		else {
			idx = (uint32_t)s->stream_slot.size();
			if (idx >= kQueueSlotsMax) {
				fail(std::string(who) + ": more than " + std::to_string(kQueueSlotsMax) + " different streams have launched on this scene");
				return nullptr;
			}
			if (idx >= kQueueSlots * (1u + (uint32_t)s->queue_more.size())) {
				int *block = nullptr;
				if (hipMalloc(&block, kQueueSlots * sizeof(int)) != hipSuccess) { fail(std::string(who) + ": hipMalloc of a block of queue counters failed"); return nullptr; }
				try { s->queue_more.push_back(block); } catch (const std::exception &) { (void)hipFree(block); fail(std::string(who) + ": out of memory"); return nullptr; }
			}
			s->stream_slot.emplace((void *)st, idx);
		}
		slot = (idx < kQueueSlots ? s->d_queue : s->queue_more[idx / kQueueSlots - 1]) + idx % kQueueSlots;
	}
	if (hipMemsetAsync(slot, 0, sizeof(int), st) != hipSuccess) { fail(std::string(who) + ": hipMemsetAsync failed"); return nullptr; }
	return slot;
}

// Persistent grid: as many workgroups as the device holds at once (waves stay until the queue is empty), fewer for
// launches too small to fill them.  The occupancy query is made once per kernel and scene.
uint32_t persistent_grid_of(rtmi_scene *s, const void *kernel, uint64_t n_items) {
	int per_cu = 0;
	{
		std::lock_guard<std::mutex> lock(s->queue_mutex);
		auto it = s->blocks_per_cu.find(kernel);
		if (it != s->blocks_per_cu.end()) per_cu = it->second;
		else {
			if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlock, s->stack_bytes) != hipSuccess || per_cu < 1) per_cu = 4;
			s->blocks_per_cu.emplace(kernel, per_cu);
		}
	}
	const uint64_t full = (uint64_t)per_cu * (uint64_t)s->n_cus;
	// RTMI_MIN_RAYS_PER_LANE (experiment, tools/quick_small_grid.py): a launch with fewer items per resident lane than this
	// gets a smaller grid — fewer waves that live longer and refill more often, at a lower occupancy
	const uint64_t min_rays = env_uint("RTMI_MIN_RAYS_PER_LANE", 0);
	const uint64_t want = min_rays ? std::max<uint64_t>((uint64_t)s->n_cus, (n_items + kBlock * min_rays - 1) / (kBlock * min_rays)) : full;
	return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::min(full, want), (n_items + kBlock - 1) / kBlock));
}
template <class K>
static uint32_t persistent_grid(rtmi_scene *s, K kernel, uint64_t n_items) {
	return persistent_grid_of(s, reinterpret_cast<const void *>(kernel), n_items);
}

template <bool ANY>
static int trace_dev(rtmi_scene *s, const rtmi_ray *d_rays, uint64_t n, rtmi_hit *d_out, uint8_t *d_occ, void *stream,
                     const char *who, const uint32_t *d_n = nullptr, uint32_t mul = 1) {
	if (check_scene(s, who)) return -1;
	if (n == 0) return 0;
	if (!d_rays || (ANY ? !d_occ : !d_out)) return fail(std::string(who) + ": NULL buffer");
	if (n > 0xff000000ull) return fail(std::string(who) + ": too many rays for one launch");
	int *slot = queue_slot(s, (hipStream_t)stream, who);
	if (!slot) return -1;
	const uint32_t grid = persistent_grid(s, k_trace_rays<ANY>, n);
	Tune tune = launch_tune(refill_threshold(), grid, n);
	// occlusion rays are often short (an AO ray with tmax = 2 rarely gets past the top of the tree): fetching their
	// successors pays only for larger groups of lanes (C4 room, 33 M AO rays: 2.21 ms at 16, 1.82 ms at 48; heavy occlusion rays prefer 16: 32 is the compromise)
	static const unsigned refill_any = std::min(64u, env_uint("RTMI_REFILL_ANY", 32));
	if (ANY) tune.thresh = (int)refill_any;
	hipLaunchKernelGGL(k_trace_rays<ANY>, dim3(grid), dim3(kBlock), s->stack_bytes, (hipStream_t)stream, s->dev, d_rays, n, d_n, mul,
	                   slot, tune, d_out, d_occ);
	return launch_check(who);
}

// for the other translation units: the same launches with the ray count read on the device (see k_trace_rays)
int closest_hit_counted(rtmi_scene *s, const rtmi_ray *d_rays, uint64_t cap, const uint32_t *d_n, uint32_t mul, rtmi_hit *d_out,
                        void *stream, const char *who) {
	return trace_dev<false>(s, d_rays, cap, d_out, nullptr, stream, who, d_n, mul);
}
int any_hit_counted(rtmi_scene *s, const rtmi_ray *d_rays, uint64_t cap, const uint32_t *d_n, uint32_t mul, uint8_t *d_occ,
                    void *stream, const char *who) {
	return trace_dev<true>(s, d_rays, cap, nullptr, d_occ, stream, who, d_n, mul);
}

static int ao_occlusion_dev(rtmi_scene *s, const double *d_normals, const rtmi_ray *d_rays, const rtmi_hit *d_hits, uint64_t n, uint32_t spp,
                            double radius, uint64_t seed, uint8_t *d_occ, void *stream) {
	const char *who = "rtmi_ao_occlusion_dev";
	if (check_scene(s, who)) return -1;
	if (n == 0 || spp == 0) return 0;
	if (!d_normals || !d_rays || !d_hits || !d_occ) return fail(std::string(who) + ": NULL argument");
	if (!(radius > 0.0)) return fail(std::string(who) + ": radius must be positive");
	if (n * spp > 0xff000000ull) return fail(std::string(who) + ": too many rays for one launch");
	if (s->n_tris == 0) { // nothing can be hit: every slot is a miss
		if (hipMemsetAsync(d_occ, 0, n * spp, (hipStream_t)stream) != hipSuccess) return fail(std::string(who) + ": hipMemsetAsync failed");
		return 0;
	}
	int *slot = queue_slot(s, (hipStream_t)stream, who);
	if (!slot) return -1;
	const uint32_t grid = persistent_grid(s, k_ao_occlusion, n * spp);
	Tune tune = launch_tune(refill_threshold(), grid, n * spp);
	// building a ray costs more than reading one, and rays of length <= radius are short: refill whole waves
	// (C4 room, 33 M rays: 2.39 / 1.97 / 1.85 / 1.81 ms at 16 / 32 / 48 / 64)
	static const unsigned refill_ao = std::min(64u, env_uint("RTMI_REFILL_AO", 64));
	tune.thresh = (int)refill_ao;
	hipLaunchKernelGGL(k_ao_occlusion, dim3(grid), dim3(kBlock), s->stack_bytes, (hipStream_t)stream, s->dev,
	                   s->d_v1_orig, d_normals, d_rays, d_hits, (uint32_t)(n * spp), spp, radius, seed, slot, tune, d_occ);
	return launch_check(who);
}

static int shadow_occlusion_dev(rtmi_scene *s, const double *d_normals, const rtmi_ray *d_rays, const rtmi_hit *d_hits, uint64_t n,
                                const double light_center[3], uint8_t *d_occ, void *stream) {
	const char *who = "rtmi_shadow_occlusion_dev";
	if (check_scene(s, who)) return -1;
	if (n == 0) return 0;
	if (!d_normals || !d_rays || !d_hits || !light_center || !d_occ) return fail(std::string(who) + ": NULL argument");
	if (n > 0xff000000ull) return fail(std::string(who) + ": too many rays for one launch");
	if (s->n_tris == 0) {
		if (hipMemsetAsync(d_occ, 0, n, (hipStream_t)stream) != hipSuccess) return fail(std::string(who) + ": hipMemsetAsync failed");
		return 0;
	}
	int *slot = queue_slot(s, (hipStream_t)stream, who);
	if (!slot) return -1;
	const uint32_t grid = persistent_grid(s, k_shadow_occlusion, n);
	Tune tune = launch_tune(refill_threshold(), grid, n);
	static const unsigned refill_any = std::min(64u, env_uint("RTMI_REFILL_ANY", 32));
	tune.thresh = (int)refill_any;
	hipLaunchKernelGGL(k_shadow_occlusion, dim3(grid), dim3(kBlock), s->stack_bytes, (hipStream_t)stream, s->dev,
	                   s->d_v1_orig, d_normals, d_rays, d_hits, (uint32_t)n, light_center[0], light_center[1], light_center[2], slot, tune, d_occ);
	return launch_check(who);
}

static int soft_shadow_dev(rtmi_scene *s, const double *d_normals, const rtmi_ray *d_rays, const rtmi_hit *d_hits, uint64_t n,
                           const double light_center[3], uint32_t n_points, uint8_t *d_contrib, double *d_factor, void *stream) {
	const char *who = "rtmi_soft_shadow_dev";
	if (check_scene(s, who)) return -1;
	if (n == 0) return 0;
	if (!d_normals || !d_rays || !d_hits || !light_center || !d_contrib || !d_factor || n_points == 0)
		return fail(std::string(who) + ": NULL argument or n_points == 0");
	if (n * n_points > 0xff000000ull) return fail(std::string(who) + ": too many rays for one launch");
	hipStream_t st = (hipStream_t)stream;
	if (s->n_tris == 0) {
		if (hipMemsetAsync(d_contrib, 0, n * n_points, st) != hipSuccess) return fail(std::string(who) + ": hipMemsetAsync failed");
	} else {
		int *slot = queue_slot(s, st, who);
		if (!slot) return -1;
		uint32_t div_x = n_points, div_y = 1; // shader.cpp:60-63
		while (div_x % 2 == 0) { div_x /= 2; div_y *= 2; }
		const uint32_t grid = persistent_grid(s, k_soft_shadow, n * n_points);
		Tune tune = launch_tune(refill_threshold(), grid, n * n_points);
		static const unsigned refill_soft = std::min(64u, env_uint("RTMI_REFILL_SOFT", 48)); // C4 room, 52 M rays: 4.76 / 4.20 / 4.00 / 4.04 ms at 16 / 32 / 48 / 64
		tune.thresh = (int)refill_soft;
		hipLaunchKernelGGL(k_soft_shadow, dim3(grid), dim3(kBlock), s->stack_bytes, st, s->dev, s->d_v1_orig,
		                   d_normals, d_rays, d_hits, (uint32_t)(n * n_points), n_points, div_x, div_y, light_center[0], light_center[1],
		                   light_center[2], slot, tune, d_contrib);
	}
	hipLaunchKernelGGL(k_soft_shadow_reduce, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, d_contrib, n, n_points, d_factor);
	return launch_check(who);
}

// host-pointer convenience wrappers: stage through device buffers, synchronous
constexpr uint64_t kSmallBatch = 4096;
template <bool ANY>
static int trace_host(rtmi_scene *s, const rtmi_ray *rays, uint64_t n, rtmi_hit *out, uint8_t *occ, const char *who) {
	if (check_scene(s, who)) return -1;
	if (n == 0) return 0;
	if (!rays || (ANY ? !occ : !out)) return fail(std::string(who) + ": NULL buffer");
	rtmi_ray *d_rays = nullptr;
	void *d_res = nullptr;
	const size_t rb = n * sizeof(rtmi_ray), ob = ANY ? n : n * sizeof(rtmi_hit);
	// Small batches (the reference's per-ray Scene::Cast left un-batched ends up here): no allocation and no staged copies —
	// the kernel reads the rays from, and writes the records to, a host-mapped buffer the scene keeps (45 -> 20 us per call).
	if (n <= kSmallBatch) {
		std::lock_guard<std::mutex> lock(s->small_mutex);
		if (!s->h_small) {
			void *h = nullptr, *d = nullptr;
			if (hipHostMalloc(&h, kSmallBatch * (sizeof(rtmi_ray) + sizeof(rtmi_hit)), hipHostMallocMapped) == hipSuccess &&
			    hipHostGetDevicePointer(&d, h, 0) == hipSuccess) { s->h_small = h; s->d_small = d; }
			else { if (h) (void)hipHostFree(h); (void)hipGetLastError(); } // fall back to the allocating path below
		}
		if (s->h_small) {
			std::memcpy(s->h_small, rays, rb);
			char *h_res = static_cast<char *>(s->h_small) + kSmallBatch * sizeof(rtmi_ray);
			char *dv_res = static_cast<char *>(s->d_small) + kSmallBatch * sizeof(rtmi_ray);
			int rc = trace_dev<ANY>(s, static_cast<const rtmi_ray *>(s->d_small), n, reinterpret_cast<rtmi_hit *>(dv_res),
			                        reinterpret_cast<uint8_t *>(dv_res), nullptr, who);
			if (!rc && hipStreamSynchronize(nullptr) != hipSuccess) rc = fail(std::string(who) + ": " + hipGetErrorString(hipGetLastError()));
			if (!rc) std::memcpy(ANY ? (void *)occ : (void *)out, h_res, ob);
			return rc;
		}
	}
	if (hipMalloc(&d_rays, rb) != hipSuccess) return fail(std::string(who) + ": hipMalloc failed");
	if (hipMalloc(&d_res, ob) != hipSuccess) { (void)hipFree(d_rays); return fail(std::string(who) + ": hipMalloc failed"); }
	int rc = 0;
	if (hipMemcpy(d_rays, rays, rb, hipMemcpyHostToDevice) != hipSuccess) rc = fail(std::string(who) + ": H2D copy failed");
	if (!rc) rc = trace_dev<ANY>(s, d_rays, n, (rtmi_hit *)d_res, (uint8_t *)d_res, nullptr, who);
	if (!rc && hipMemcpy(ANY ? (void *)occ : (void *)out, d_res, ob, hipMemcpyDeviceToHost) != hipSuccess)
		rc = fail(std::string(who) + ": D2H copy failed: " + hipGetErrorString(hipGetLastError()));
	(void)hipFree(d_rays);
	(void)hipFree(d_res);
	return rc;
}

} // namespace rtmi

using namespace rtmi;

extern "C" {

const char *rtmi_last_error(void) { return g_err.c_str(); }

int rtmi_device_count(void) {
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
	return n;
}

rtmi_scene *rtmi_scene_create(const double *tri_xyz, const uint32_t *tri_object, uint64_t n_tris) {
	try {
		return scene_create(tri_xyz, tri_object, n_tris);
	} catch (const std::exception &e) { // std::bad_alloc of the host build at 2^27 triangles, std::system_error of its threads
		fail(std::string("rtmi_scene_create: ") + e.what());
		return nullptr;
	}
}

int rtmi_scene_check(const rtmi_scene *s) { return check_scene(s, "rtmi_scene_check"); }

void rtmi_scene_destroy(rtmi_scene *s) {
	if (!s) return;
	if (s->dev.nodes) (void)hipFree((void *)s->dev.nodes);
	if (s->dev.wnodes) (void)hipFree((void *)s->dev.wnodes);
	if (s->dev.tris) (void)hipFree((void *)s->dev.tris);
	if (s->d_v1_orig) (void)hipFree(s->d_v1_orig);
	if (s->d_queue) (void)hipFree(s->d_queue);
	for (int *b : s->queue_more) (void)hipFree(b);
	if (s->h_error) (void)hipHostFree(s->h_error);
	if (s->h_small) (void)hipHostFree(s->h_small); // the small-batch buffer of the host-pointer calls (trace_host)
	delete s;
}

int rtmi_scene_get_info(const rtmi_scene *s, rtmi_scene_info *out) {
	if (!s || !out) return fail("rtmi_scene_get_info: NULL argument");
	*out = s->info;
	return 0;
}

int rtmi_closest_hit(rtmi_scene *s, const rtmi_ray *rays, uint64_t n, rtmi_hit *out) {
	return trace_host<false>(s, rays, n, out, nullptr, "rtmi_closest_hit");
}
int rtmi_closest_hit_dev(rtmi_scene *s, const rtmi_ray *d_rays, uint64_t n, rtmi_hit *d_out, void *stream) {
	return trace_dev<false>(s, d_rays, n, d_out, nullptr, stream, "rtmi_closest_hit_dev");
}
int rtmi_any_hit(rtmi_scene *s, const rtmi_ray *rays, uint64_t n, uint8_t *occluded) {
	return trace_host<true>(s, rays, n, nullptr, occluded, "rtmi_any_hit");
}
int rtmi_shadow_occlusion_dev(rtmi_scene *s, const double *d_normals, const rtmi_ray *d_rays, const rtmi_hit *d_hits, uint64_t n,
                              const double light_center[3], uint8_t *d_occluded, void *stream) {
	return shadow_occlusion_dev(s, d_normals, d_rays, d_hits, n, light_center, d_occluded, stream);
}

int rtmi_soft_shadow_dev(rtmi_scene *s, const double *d_normals, const rtmi_ray *d_rays, const rtmi_hit *d_hits, uint64_t n,
                         const double light_center[3], uint32_t n_points, uint8_t *d_contrib, double *d_factor, void *stream) {
	return soft_shadow_dev(s, d_normals, d_rays, d_hits, n, light_center, n_points, d_contrib, d_factor, stream);
}

static int debug_check_hierarchy(const double *tri_xyz, uint64_t n_tris, uint64_t stats[4]);
int rtmi_debug_check_hierarchy(const double *tri_xyz, uint64_t n_tris, uint64_t stats[4]) {
	try {
		return debug_check_hierarchy(tri_xyz, n_tris, stats);
	} catch (const std::exception &e) {
		return fail(std::string("rtmi_debug_check_hierarchy: ") + e.what());
	}
}
static int debug_check_hierarchy(const double *tri_xyz, uint64_t n_tris, uint64_t stats[4]) {
	const char *who = "rtmi_debug_check_hierarchy";
	if ((n_tris > 0 && !tri_xyz) || !stats) return fail(std::string(who) + ": NULL argument");
	if (n_tris >= kMaxFirst) return fail(std::string(who) + ": more than 2^27-1 triangles");
	const unsigned threads = std::min(env_uint("RTMI_BUILD_THREADS", std::max(1u, std::thread::hardware_concurrency())), 16u);
	Bvh2 bvh = build_bvh2(tri_xyz, n_tris, 1, kStackDepth - 2, threads);
	Wide4 wide;
	std::string err;
	if (build_wide(bvh, wide, err, threads)) return fail(std::string(who) + ": " + err);
	stats[0] = check_wide(bvh, wide);
	stats[1] = wide.nodes.size();
	stats[2] = wide.stack_need;
	stats[3] = bvh.depth;
	return 0;
}

// test hook: the shared-reciprocal vector division of rtmi_device.hpp beside the compiler's plain divisions
__global__ void k_debug_div3(const double *__restrict__ num3, const double *__restrict__ den, uint64_t n, double *__restrict__ shared3,
                             double *__restrict__ plain3) {
	const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= n) return;
	const D3 a = d3(num3[3 * k], num3[3 * k + 1], num3[3 * k + 2]);
	const double c = den[k];
	const D3 q = div3(a, c);
	shared3[3 * k] = q.x; shared3[3 * k + 1] = q.y; shared3[3 * k + 2] = q.z;
	plain3[3 * k] = a.x / c; plain3[3 * k + 1] = a.y / c; plain3[3 * k + 2] = a.z / c;
}
#ifdef RTMI_TAIL_DIAG
extern "C" int rtmi_debug_tail_read(unsigned long long *host_out) { // diagnostic builds only (tools/tail_diag.py); not in rtmi.h
	return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(rtmi::g_tail_diag), sizeof(unsigned long long) * 5 * 8192) == hipSuccess ? 0 : -1;
}
#endif
__global__ void k_debug_unit(const double *__restrict__ vec3, uint64_t n, double *__restrict__ fast3, double *__restrict__ plain3) {
	const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const bool in = k < n; // (every lane runs the wave-level vote of unit_of_len2)
	const D3 a = in ? d3(vec3[3 * k], vec3[3 * k + 1], vec3[3 * k + 2]) : d3(1.0, 2.0, 3.0);
	const double d2 = len2(a);
	const D3 q = unit_of_len2(a, d2);
	const double s = sqrt(d2);
	if (in) {
		fast3[3 * k] = q.x; fast3[3 * k + 1] = q.y; fast3[3 * k + 2] = q.z;
		plain3[3 * k] = a.x / s; plain3[3 * k + 1] = a.y / s; plain3[3 * k + 2] = a.z / s;
	}
}

int rtmi_debug_div3_dev(const double *d_num3, const double *d_den, uint64_t n, double *d_shared3, double *d_plain3, void *stream) {
	const char *who = "rtmi_debug_div3_dev";
	if (n == 0) return 0;
	if (!d_num3 || !d_den || !d_shared3 || !d_plain3) return fail(std::string(who) + ": NULL argument");
	if ((n + 255) / 256 > 0x7fffffffull) return fail(std::string(who) + ": too many elements");
	hipLaunchKernelGGL(k_debug_div3, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_num3, d_den, n, d_shared3, d_plain3);
	return launch_check(who);
}

int rtmi_debug_unit_dev(const double *d_vec3, uint64_t n, double *d_fast3, double *d_plain3, void *stream) {
	const char *who = "rtmi_debug_unit_dev";
	if (!d_vec3 || !d_fast3 || !d_plain3) return fail(std::string(who) + ": NULL argument");
	if (n == 0) return 0;
	if ((n + 255) / 256 > 0x7fffffffull) return fail(std::string(who) + ": too many elements");
	hipLaunchKernelGGL(k_debug_unit, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_vec3, n, d_fast3, d_plain3);
	return launch_check(who);
}

int rtmi_any_hit_dev(rtmi_scene *s, const rtmi_ray *d_rays, uint64_t n, uint8_t *d_occluded, void *stream) {
	return trace_dev<true>(s, d_rays, n, nullptr, d_occluded, stream, "rtmi_any_hit_dev");
}

int rtmi_ao_occlusion_dev(rtmi_scene *s, const double *d_normals, const rtmi_ray *d_rays, const rtmi_hit *d_hits, uint64_t n, uint32_t spp,
                          double radius, uint64_t seed, uint8_t *d_occluded, void *stream) {
	return ao_occlusion_dev(s, d_normals, d_rays, d_hits, n, spp, radius, seed, d_occluded, stream);
}

int rtmi_triangle_intersection(uint64_t n, const double *o, const double *d, const double *tri, double *t, double *u,
                               double *v) {
	const char *who = "rtmi_triangle_intersection";
	if (n == 0) return 0;
	if (!o || !d || !tri || !t || !u || !v) return fail(std::string(who) + ": NULL buffer");
	if (n > (uint64_t)0x7fffffff * 256) return fail(std::string(who) + ": too many pairs for one launch");
	double *buf = nullptr; // o[3n] d[3n] tri[9n] t[n] u[n] v[n]
	if (hipMalloc(&buf, sizeof(double) * 18 * n) != hipSuccess) return fail(std::string(who) + ": hipMalloc failed");
	double *d_o = buf, *d_d = buf + 3 * n, *d_tri = buf + 6 * n, *d_t = buf + 15 * n, *d_u = buf + 16 * n, *d_v = buf + 17 * n;
	bool ok = hipMemcpy(d_o, o, 24 * n, hipMemcpyHostToDevice) == hipSuccess &&
	          hipMemcpy(d_d, d, 24 * n, hipMemcpyHostToDevice) == hipSuccess &&
	          hipMemcpy(d_tri, tri, 72 * n, hipMemcpyHostToDevice) == hipSuccess &&
	          hipMemcpy(d_u, u, 8 * n, hipMemcpyHostToDevice) == hipSuccess &&
	          hipMemcpy(d_v, v, 8 * n, hipMemcpyHostToDevice) == hipSuccess;
	int rc = ok ? 0 : fail(std::string(who) + ": H2D copy failed");
	if (!rc) {
		hipLaunchKernelGGL(k_triangle_intersection, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, nullptr, n, d_o, d_d, d_tri,
		                   d_t, d_u, d_v);
		rc = launch_check(who);
	}
	if (!rc && (hipMemcpy(t, d_t, 8 * n, hipMemcpyDeviceToHost) != hipSuccess ||
	            hipMemcpy(u, d_u, 8 * n, hipMemcpyDeviceToHost) != hipSuccess ||
	            hipMemcpy(v, d_v, 8 * n, hipMemcpyDeviceToHost) != hipSuccess))
		rc = fail(std::string(who) + ": D2H copy failed: " + hipGetErrorString(hipGetLastError()));
	(void)hipFree(buf);
	return rc;
}

// Render2/rasterizer.cpp:40-56 with the reference's Vec3 operators (vec3.h) restated on the host.
int rtmi_camera_setup(const double eye[3], const double lookat[3], const double up[3], double vpdist,
                      const double x_win[2], const double y_win[2], uint32_t x_res, uint32_t y_res, rtmi_camera *out) {
	if (!eye || !lookat || !up || !x_win || !y_win || !out) return fail("rtmi_camera_setup: NULL argument");
	if (x_res == 0 || y_res == 0) return fail("rtmi_camera_setup: zero resolution");
	struct V { double x, y, z; };
	auto sub = [](V a, V b) { return V{a.x - b.x, a.y - b.y, a.z - b.z}; };
	auto add = [](V a, V b) { return V{a.x + b.x, a.y + b.y, a.z + b.z}; };
	auto scl = [](double s, V a) { return V{s * a.x, s * a.y, s * a.z}; };
	auto dotp = [](V a, V b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); };
	auto divs = [](V a, double c) { return V{a.x / c, a.y / c, a.z / c}; };
	auto l2 = [](V a) { return a.x * a.x + a.y * a.y + a.z * a.z; };
	auto cross = [](V a, V b) { return V{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; };
	auto unitv = [&](V a) { double d = l2(a); return d > 0.0 ? divs(a, std::sqrt(d)) : V{0, 0, 0}; };
	const V E{eye[0], eye[1], eye[2]}, L{lookat[0], lookat[1], lookat[2]}, Up{up[0], up[1], up[2]};
	double width = x_win[1] - x_win[0], height = y_win[1] - y_win[0]; // interval.h:15-20
	if (!(width > 0.0)) width = 0.0;
	if (!(height > 0.0)) height = 0.0;
	const V G = unitv(sub(L, E));
	V upG = Up; // vec3.h:70-75, operator/(Vec3,Vec3): remove the component parallel to G
	{
		double x = l2(G);
		if (x > 0.0) upG = sub(Up, scl(dotp(Up, G) / x, G));
	}
	const V U = unitv(upG);
	const V R = unitv(cross(G, U));
	const V O = add(add(scl(vpdist, G), scl(x_win[0], R)), scl(y_win[1], U));
	const V dR = divs(scl(width, R), (double)x_res);
	const V dU = divs(scl(height, U), (double)y_res);
	out->eye[0] = E.x; out->eye[1] = E.y; out->eye[2] = E.z;
	out->O[0] = O.x; out->O[1] = O.y; out->O[2] = O.z;
	out->dR[0] = dR.x; out->dR[1] = dR.y; out->dR[2] = dR.z;
	out->dU[0] = dU.x; out->dU[1] = dU.y; out->dU[2] = dU.z;
	out->x_res = x_res;
	out->y_res = y_res;
	return 0;
}

// tiles -> workgroup blocks; returns the grid size (0: nothing to do), -1 on error
static int64_t primary_grid(const rtmi_camera *cam, uint32_t tile_w, uint32_t tile_h, uint32_t tile_first, uint32_t tile_stride,
                            PrimaryGrid &g, const char *who) {
	if (cam->x_res == 0 || cam->y_res == 0) return fail(std::string(who) + ": zero resolution");
	if (tile_w == 0 || tile_h == 0) { tile_w = cam->x_res; tile_h = cam->y_res; }
	if (tile_stride == 0) return fail(std::string(who) + ": tile_stride must be >= 1");
	const uint32_t tiles_x = (cam->x_res + tile_w - 1) / tile_w, tiles_y = (cam->y_res + tile_h - 1) / tile_h;
	const uint64_t n_tiles = (uint64_t)tiles_x * tiles_y;
	if (tile_first >= n_tiles) return 0;
	const uint64_t local = (n_tiles - tile_first + tile_stride - 1) / tile_stride;
	g.tile_w = tile_w; g.tile_h = tile_h; g.tiles_x = tiles_x; g.tile_first = tile_first; g.tile_stride = tile_stride;
	g.n_tiles = (uint32_t)n_tiles;
	g.blk_w = std::min(64u, (tile_w + 7) / 8 * 8);
	g.blk_h = std::min(std::max(8u, env_uint("RTMI_BLK_H", 32) / 8 * 8), (tile_h + 7) / 8 * 8);
	g.blocks_x = (tile_w + g.blk_w - 1) / g.blk_w;
	g.blocks_per_tile = g.blocks_x * ((tile_h + g.blk_h - 1) / g.blk_h);
	g.d_per_block = make_fastdiv(g.blk_w * g.blk_h);
	g.d_blocks_per_tile = make_fastdiv(g.blocks_per_tile);
	g.d_tiles_x = make_fastdiv(g.tiles_x);
	g.d_blocks_x = make_fastdiv(g.blocks_x);
	g.d_sq_x = make_fastdiv(g.blk_w >> 3);
	const uint64_t grid = local * g.blocks_per_tile;
	if (grid > 0x7fffffffull) return fail(std::string(who) + ": grid too large");
	return (int64_t)grid;
}

int rtmi_primary_closest_hit_dev(rtmi_scene *s, const rtmi_camera *cam, double sx, double sy, uint32_t tile_w,
                                 uint32_t tile_h, uint32_t tile_first, uint32_t tile_stride, rtmi_hit *d_out,
                                 void *stream) {
	const char *who = "rtmi_primary_closest_hit_dev";
	if (check_scene(s, who)) return -1;
	if (!cam || !d_out) return fail(std::string(who) + ": NULL argument");
	PrimaryGrid g;
	const int64_t grid = primary_grid(cam, tile_w, tile_h, tile_first, tile_stride, g, who);
	if (grid <= 0) return (int)grid;
	const uint64_t n_items = (uint64_t)grid * g.blk_w * g.blk_h;
	if (n_items > 0x7fffffffull) return fail(std::string(who) + ": too many pixels for one launch"); // (item indices below 2^31: FastDiv)
	int *slot = queue_slot(s, (hipStream_t)stream, who);
	if (!slot) return -1;
	const uint32_t pgrid = persistent_grid(s, k_primary_closest, n_items);
	hipLaunchKernelGGL(k_primary_closest, dim3(pgrid), dim3(kBlock), s->stack_bytes, (hipStream_t)stream, s->dev, slot, (uint32_t)n_items, *cam, sx,
	                   sy, g, launch_tune(refill_threshold(), pgrid, n_items), d_out);
	return launch_check(who);
}

int rtmi_primary_closest_hit_samples_dev(rtmi_scene *s, const rtmi_camera *cam, uint32_t tile_w, uint32_t tile_h,
                                         uint32_t tile_stride, const rtmi_primary_sample *samples, uint32_t n_samples,
                                         void *stream) {
	const char *who = "rtmi_primary_closest_hit_samples_dev";
	if (check_scene(s, who)) return -1;
	if (!cam || (!samples && n_samples)) return fail(std::string(who) + ": NULL argument");
	if (n_samples > RTMI_MAX_SAMPLES_PER_LAUNCH)
		return fail(std::string(who) + ": at most " + std::to_string(RTMI_MAX_SAMPLES_PER_LAUNCH) + " samples per launch");
	if (n_samples == 0) return 0;
	// every sample gets the item range of the one with the most tiles; a sample's items beyond its last tile hold no ray
	SampleSet set;
	memset(&set, 0, sizeof set);
	uint32_t lowest = samples[0].tile_first;
	for (uint32_t k = 0; k < n_samples; ++k) {
		if (!samples[k].d_out) return fail(std::string(who) + ": sample without a frame");
		set.s[k] = samples[k];
		lowest = std::min(lowest, samples[k].tile_first);
	}
	PrimaryGrid g;
	const int64_t grid = primary_grid(cam, tile_w, tile_h, lowest, tile_stride, g, who);
	if (grid <= 0) return (int)grid;
	const uint64_t per_sample = (uint64_t)grid * g.blk_w * g.blk_h, n_items = per_sample * n_samples;
	if (n_items > 0x7fffffffull) return fail(std::string(who) + ": too many pixels for one launch"); // (item indices below 2^31: FastDiv)
	set.n = n_samples;
	set.items_per_sample = (uint32_t)per_sample;
	set.d_items_per_sample = make_fastdiv((uint32_t)per_sample);
	int *slot = queue_slot(s, (hipStream_t)stream, who);
	if (!slot) return -1;
	const uint32_t pgrid = persistent_grid(s, k_primary_closest_samples, n_items);
	hipLaunchKernelGGL(k_primary_closest_samples, dim3(pgrid), dim3(kBlock), s->stack_bytes, (hipStream_t)stream, s->dev, slot,
	                   (uint32_t)n_items, *cam, g, launch_tune(refill_threshold(), pgrid, n_items), set);
	return launch_check(who);
}

int rtmi_primary_work_dev(rtmi_scene *s, const rtmi_camera *cam, double sx, double sy, uint64_t *d_counts, void *stream) {
	const char *who = "rtmi_primary_work_dev";
	if (check_scene(s, who)) return -1;
	if (!cam || !d_counts) return fail(std::string(who) + ": NULL argument");
	PrimaryGrid g;
	const int64_t grid = primary_grid(cam, 64, 64, 0, 1, g, who);
	if (grid <= 0) return (int)grid;
	const uint64_t n_items = (uint64_t)grid * g.blk_w * g.blk_h;
	if (n_items > 0x7fffffffull) return fail(std::string(who) + ": too many pixels for one launch"); // (item indices below 2^31: FastDiv)
	int *slot = queue_slot(s, (hipStream_t)stream, who);
	if (!slot) return -1;
	const uint32_t pgrid = persistent_grid(s, k_primary_work, n_items);
	hipLaunchKernelGGL(k_primary_work, dim3(pgrid), dim3(kBlock), s->stack_bytes, (hipStream_t)stream, s->dev, slot, (uint32_t)n_items, *cam, sx, sy,
	                   g, launch_tune(refill_threshold(), pgrid, n_items), (unsigned long long *)d_counts);
	return launch_check(who);
}

int rtmi_primary_rays_dev(const rtmi_camera *cam, double sx, double sy, rtmi_ray *d_rays, void *stream) {
	if (!cam || !d_rays) return fail("rtmi_primary_rays_dev: NULL argument");
	const uint64_t n = (uint64_t)cam->x_res * cam->y_res;
	if (n == 0) return 0;
	hipLaunchKernelGGL(k_primary_rays, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *cam, sx, sy, d_rays);
	return launch_check("rtmi_primary_rays_dev");
}

} // extern "C"
