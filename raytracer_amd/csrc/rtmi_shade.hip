// librtmi — Scene::Trace + Shader::Shade as a wavefront on the device: the caller of the ray queries, so that a whole
// frame (primary hits, shadow and ambient-occlusion rays, reflected and refracted generations, colours, ToneMap) never
// leaves HBM.  Reference: Render2/scene.cpp:127-155 (Trace), Render2/shader.cpp:395-645 (Shade, the branch main.cpp
// runs: scene.radiosity == NULL), :105-138 (generateNormalShadows), :140-232 (findOcclusionNew),
// Render2/rasterizer.cpp:79-88 (ToneMap), :219-233 (the pixel loop).  All colour arithmetic is IEEE double in the
// reference's operation order; pow/exp/sin/cos are the device library's, so colours agree with the reference to a
// few ulp, not to the bit (hit ids, shadow and occlusion bits are exact).  gfx950 only.
//
// The reference recurses per pixel; here one generation of the ray tree is one batch:
//   closest hit -> k_shade_pre (hit point, shadow ray, refraction children) -> any hit (shadow) -> k_ao_list
//   (+ AO frames) -> k_ao_trace (AO rays built in the any-hit kernel) -> k_shade_post (colour, reflection child) -> next generation ... -> k_fold
// Generations are COMPACT: a node reserves slots for its children from a device-side counter (one atomic per wave),
// AO rays exist only for the nodes that will use the answer, and every launch reads its item count on the device, so a
// frame is queued without the host ever waiting.  Work the reference does but whose result it never uses is skipped:
// AO where the point is shadowed or the material's ambient is 0, the local colour of refracting materials, a reflected
// ray whose weight is 0.
#include <cmath>
#include <cstring>
#include <vector>

#include "rtmi_host.hpp"

struct rtmi_shader {
	rtmi_scene *scene;
	double *d_normals;
	rtmi_material *d_mats;
	rtmi_light *d_lights;
	uint32_t n_objects, n_lights;
	bool any_translucent, any_ambient;
	char *ws;        // grow-only workspace
	size_t ws_bytes;
};

namespace rtmi {

struct ShadeDev {
	const double *v1_orig, *normals;
	const rtmi_material *mats;
	const rtmi_light *lights;
	uint32_t n_lights;
};

enum : uint32_t { kDead = 0, kFinal = 1, kOpaque = 2, kReflect = 3, kRefract = 4, kTir = 5 };
constexpr uint32_t kNone = 0xFFFFFFFFu;
struct Node { // 72 bytes per slot
	double c[3];    // colour (local until folded, then final)
	double coef[3]; // reflectivity (kReflect) or translucency weights (kRefract, kTir)
	double beer;    // Schlick weight of the reflected child (kRefract)
	uint32_t kind;
	uint32_t child;    // first child's slot in the next generation (kReflect, kTir: 1 child; kRefract: 2)
	uint32_t rootslot; // slot of the generation-1 ray this node descends from -> root[] = its ray index
	uint32_t path;     // 1 = that ray itself, 2p = reflected child of p, 2p+1 = refracted child of p
};

constexpr double kEpsilon = 1.0e-5; // Render2/base.h:57
__device__ __forceinline__ D3 background() { return d3(0.15, 0.25, 0.35); } // scene.cpp:21
__device__ __forceinline__ D3 termination() { return d3(0.0, 0.0, 0.0); }   // scene.cpp:22
__device__ __forceinline__ D3 ld(const double *p) { return d3(p[0], p[1], p[2]); }
__device__ __forceinline__ void st(double *p, D3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
__device__ __forceinline__ D3 cmul(D3 a, D3 b) { return d3(a.x * b.x, a.y * b.y, a.z * b.z); } // color.h:43
__device__ __forceinline__ double max2(double x, double y) { return x >= y ? x : y; }         // util.h:26
__device__ __forceinline__ double min3(double x, double y, double z) {                       // util.cpp:7
	if (x <= y && x <= z) return x;
	if (y <= z) return y;
	return z;
}
// one out-of-line copy of the library's pow: eight inlined copies cost k_shade_post its registers
__device__ __attribute__((noinline)) double pow_ref(double x, double y) { return pow(x, y); }
__device__ __attribute__((noinline)) double exp_ref(double x) { return exp(x); }
__device__ __forceinline__ bool refracts(const rtmi_material &m) { // shader.cpp:542
	return m.translucency[2] > 0.0 && m.translucency[1] > 0.0 && m.translucency[0] > 0.0;
}
__device__ __forceinline__ bool nonzero(const double *c) { return c[0] != 0.0 || c[1] != 0.0 || c[2] != 0.0; }

// Every thread of the workgroup calls this (converged code, block-uniform loops) with the number of slots it needs
// (0, 1 or 2); returns the thread's first slot.  One atomic per workgroup: atomics that return a value serialise on one
// address (about 8 ns each on this part), so the reserving kernels run 1024-thread workgroups.
constexpr uint32_t kReserveBlock = 1024;
__device__ __forceinline__ uint32_t reserve(uint32_t *counter, uint32_t need) {
	__shared__ uint32_t s_tot[kReserveBlock / 64];
	__shared__ uint32_t s_base;
	const unsigned long long b1 = __ballot(need >= 1), b2 = __ballot(need >= 2);
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
	if (lane == 0) s_tot[wave] = (uint32_t)(__popcll(b1) + __popcll(b2));
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t sum = 0;
		for (uint32_t w = 0; w < waves; ++w) sum += s_tot[w];
		s_base = sum ? atomicAdd(counter, sum) : 0u;
	}
	__syncthreads();
	uint32_t at = s_base;
	for (uint32_t w = 0; w < wave; ++w) at += s_tot[w];
	const unsigned long long lt = (1ull << lane) - 1ull;
	at += (uint32_t)(__popcll(b1 & lt) + __popcll(b2 & lt));
	__syncthreads(); // s_tot / s_base are rewritten by the next step
	return at;
}

// Elementwise kernels run a fixed grid and stride over the items; the item count is read on the device.  The loop
// bound is uniform over the workgroup (reserve() synchronises it); `in` tells a thread whether its item exists.
#define RTMI_FOR_ITEMS(k, in, count)                                                                          \
	for (uint64_t k##_b = (uint64_t)blockIdx.x * blockDim.x, k = k##_b + threadIdx.x; k##_b < (count);        \
	     k##_b += (uint64_t)gridDim.x * blockDim.x, k += (uint64_t)gridDim.x * blockDim.x)                      \
		if (const bool in = k < (count); true)

// Generation-1 rays of a frame, rasterizer.cpp:230.  Items enumerate this rank's tiles (tile_first, + tile_stride, ...),
// 8×8-pixel squares inside a tile when its sides allow; items that fall outside the frame are dead slots.
struct FrameGrid { uint32_t tile_w, tile_h, tiles_x, tile_first, tile_stride; };
__device__ __forceinline__ bool frame_pixel(const FrameGrid &g, const rtmi_camera &cam, uint64_t item, uint32_t &i, uint32_t &j) {
	const uint32_t per_tile = g.tile_w * g.tile_h;
	const uint32_t tile = g.tile_first + (uint32_t)(item / per_tile) * g.tile_stride, k = (uint32_t)(item % per_tile);
	uint32_t lx, ly;
	if (((g.tile_w | g.tile_h) & 7u) == 0) {
		const uint32_t sq = k >> 6, in = k & 63u, sq_x = g.tile_w >> 3;
		lx = (sq % sq_x) * 8 + (in & 7u);
		ly = (sq / sq_x) * 8 + (in >> 3);
	} else {
		lx = k % g.tile_w;
		ly = k / g.tile_w;
	}
	j = (tile % g.tiles_x) * g.tile_w + lx;
	i = (tile / g.tiles_x) * g.tile_h + ly;
	return j < cam.x_res && i < cam.y_res;
}
// counts[0] = n, every other counter = 0
__global__ void k_set_counts(uint32_t *counts, uint32_t n_counters, uint32_t n) {
	const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k < n_counters) counts[k] = k == 0 ? n : 0u;
}
// Generation-1 rays of a frame.  Item = pixel item · S + sample; S = 1 (Serial_Normal_Raster, rasterizer.cpp:230),
// 4 (Anti_Aliasing, :327-343) or lens_dim² (Depth_Of_Field_Effect, :378-395), samples in the reference's loop order.
__global__ void __launch_bounds__(256) k_frame_rays(rtmi_camera cam, rtmi_raster ras, uint32_t S, FrameGrid g, uint64_t item0, uint64_t n,
                                                    rtmi_ray *__restrict__ rays, uint64_t *__restrict__ root) {
	RTMI_FOR_ITEMS(k, in, n) {
		if (!in) continue;
		uint32_t i, j;
		const uint64_t item = item0 + k;
		const uint32_t s = (uint32_t)(item % S);
		if (frame_pixel(g, cam, item / S, i, j)) {
			const D3 O = ld(cam.O), dR = ld(cam.dR), dU = ld(cam.dU);
			D3 org = ld(cam.eye), dir;
			if (ras.mode == RTMI_RASTER_ANTI_ALIASING) {
				const float fragmenty = (s >> 1) ? 3.f : 1.f, fragmentx = (s & 1u) ? 3.f : 1.f; // :327-329
				dir = unit((O + ((j + (fragmentx * 0.25)) * dR)) - ((i + (fragmenty * 0.25)) * dU)); // :335
			} else if (ras.mode == RTMI_RASTER_DEPTH_OF_FIELD) {
				const int grad = (int)(ras.lens_dim / 2);                                     // :64
				const int yLens = grad - (int)(s / ras.lens_dim), xLens = -grad + (int)(s % ras.lens_dim); // :378-380
				org = (ld(cam.eye) + ((double)xLens * ld(ras.x_gradation))) + ((double)yLens * ld(ras.y_gradation)); // :383
				const D3 GLens = unit(ld(ras.lookat) - org);                                   // :384
				const D3 OL = ((ras.vpdist * GLens) + (ras.xmin * ld(ras.R))) + (ras.ymax * ld(ras.U)); // :385
				dir = unit((OL + ((j + 0.5) * dR)) - ((i + 0.5) * dU));                       // :386-388
			} else {
				dir = unit((O + (j + ras.sx) * dR) - (i + ras.sy) * dU);
			}
			set_ray(rays[k], org, dir, RTMI_INFINITY);
			root[k] = ((uint64_t)i * cam.x_res + j) * S + s;
		} else {
			inactive(rays[k]);
			root[k] = ~0ull;
		}
	}
}
__global__ void k_batch_roots(uint64_t first, uint64_t n, uint64_t *__restrict__ root) {
	RTMI_FOR_ITEMS(k, in, n) {
		if (in) root[k] = first + k;
	}
}

// Everything of Shade that needs no secondary-ray answer: classify the slot, refraction (shader.cpp:542-641), the
// shadow ray (:116-127).  `last`: the children would be generation max_tree_depth+1, for which Trace returns the
// termination colour without casting (scene.cpp:134-137), so they are folded in at once.
__global__ void __launch_bounds__(kReserveBlock) __attribute__((amdgpu_waves_per_eu(4, 4))) k_shade_pre(ShadeDev S, const rtmi_ray *__restrict__ rays, const rtmi_hit *__restrict__ hits,
                            const uint32_t *__restrict__ d_count, int first, int last, int radiosity_branch, Node *__restrict__ nodes,
                            rtmi_ray *__restrict__ shadow, uint8_t *__restrict__ wants_ao, uint32_t *__restrict__ kid_count,
                            rtmi_ray *__restrict__ kid_rays, Node *__restrict__ kid_nodes) {
	const uint64_t count = *d_count;
	RTMI_FOR_ITEMS(k, in, count) {
		// the node and the rays stay in registers until they are stored (no local structs: they would live in scratch)
		D3 ndc = d3(0, 0, 0), ndcoef = d3(0, 0, 0);
		double ndbeer = 0.0;
		uint32_t ndkind = kDead;
		const uint32_t rootslot = first ? (uint32_t)k : (in ? nodes[k].rootslot : 0u);
		const uint32_t path = first ? 1u : (in ? nodes[k].path : 0u);
		D3 sh_o = d3(0, 0, 0), sh_d = d3(0, 0, 0), c0_o = d3(0, 0, 0), c0_d = d3(0, 0, 0), c1_o = d3(0, 0, 0), c1_d = d3(0, 0, 0);
		bool has_shadow = false;
		uint32_t need = 0;
		uint8_t ao = 0; // an AO answer would be used: opaque, ambient != 0 (and not shadowed: k_ao_list)
		if (in) {
			const rtmi_ray &r = rays[k];
			if (r.reserved == 0.0) {
				const rtmi_hit h = hits[k];
				if (h.tri == RTMI_MISS) {
					ndkind = kFinal;
					ndc = background(); // scene.cpp:151
				} else {
					const rtmi_material &m = S.mats[h.object];
					const D3 Ns = ld(S.normals + 3 * (size_t)h.tri); // hit.normal = triNormal, object.cpp:17
					const D3 P = hit_point(r, S.v1_orig + 3 * (size_t)h.tri, Ns);
					if (radiosity_branch) {
						// shader.cpp:431-478 as shipped: radiosityAmbient (:433) and occlusion (:434) are used by nothing (:457-465
						// are comments), so the hit's colour is the specular sum over the lights, returned at :477
						const D3 N = unit(Ns);          // :416
						const D3 E = unit(ld(r.o) - P); // :417
						const D3 specular = ld(m.specular);
						D3 color = d3(0, 0, 0);         // :427-429
						for (uint32_t i = 0; i < S.n_lights; ++i) {
							const D3 emission = ld(S.lights[i].emission);
							const D3 L = unit(ld(S.lights[i].center) - P);      // :453
							const D3 RR = (-1.0 * L) + ((2.0 * dot(L, N)) * N); // :467
							color = color + cmul(emission, pow_ref(max2(0.0, dot(E, RR)), m.phong_exp) * specular); // :472
						}
						ndkind = kFinal;
						ndc = color;
					} else if (refracts(m)) {
						const double kk = m.ref_index;
						const D3 E = unit(ld(r.o) - P); // :417
						D3 transl = ld(m.translucency);
						const D3 mE = -1.0 * E, mN = -1.0 * Ns;
						D3 R, tRefr = d3(0, 0, 0);
						double refractionCoef = 0;
						bool tir = false;
						if (dot(E, Ns) < 0.0) R = (-1.0 * E) + ((2.0 * dot(E, mN)) * mN); // :571-574
						else R = (-1.0 * E) + ((2.0 * dot(E, Ns)) * Ns);
						if (dot(mE, Ns) < 0.0) { // :576-588
							const double dn = dot(mE, Ns);
							const D3 cosRefract = (mE - (dn * Ns)) / kk;
							const double cosSq = 1 - ((1 - (dn * dn)) / (kk * kk));
							tRefr = cosRefract - (sqrt(cosSq) * Ns);
							refractionCoef = dot(E, Ns);
							transl = d3(1, 1, 1);
						} else { // :589-625
							transl = d3(exp_ref(-1.0 * transl.x * h.t), exp_ref(-1.0 * transl.y * h.t), exp_ref(-1.0 * transl.z * h.t));
							const double dn = dot(mE, mN);
							const D3 cosRefract = (mE - (dn * mN)) / kk;
							const double cosSq = 1 - ((1 - (dn * dn)) / (kk * kk));
							if (cosSq < 0.0) tir = true; // :608-615
							else {
								tRefr = cosRefract - (sqrt(cosSq) * mN);
								refractionCoef = dot(tRefr, Ns);
							}
						}
						ndcoef = transl;
						if (tir) {
							if (last) { ndkind = kFinal; ndc = cmul(transl, termination()); }
							else { ndkind = kTir; need = 1; c0_o = P + (kEpsilon * R); c0_d = R; }
						} else {
							const double beerR0 = ((1 - kk) * (1 - kk)) / ((1 + kk) * (1 + kk));         // :627
							const double beerR = beerR0 + (1 - beerR0) * pow_ref(1 - refractionCoef, 5.0); // :628
							ndbeer = beerR;
							if (last) {
								ndkind = kFinal;
								ndc = cmul(transl, (beerR * termination()) + ((1 - beerR) * termination())); // :640
							} else {
								ndkind = kRefract;
								need = 2;
								c0_o = P + (kEpsilon * R); c0_d = R;         // :630-633
								c1_o = P + (kEpsilon * tRefr); c1_d = tRefr; // :635-638
							}
						}
					} else {
						ndkind = kOpaque; // colour and reflection: k_shade_post.  Until then the colour slot carries the hit point
						ndc = P;          // for k_ao_list and k_shade_post (the same doubles they would recompute)
						ao = nonzero(m.ambient) ? 1 : 0;
						has_shadow = true;
						sh_o = P + (0.01 * Ns); sh_d = unit(ld(S.lights[0].center) - P); // :122,127
					}
				}
			}
		}
		const uint32_t at = reserve(kid_count, need);
		if (need) {
			set_ray(kid_rays[at], c0_o, c0_d, RTMI_INFINITY);
			kid_nodes[at].rootslot = rootslot; kid_nodes[at].path = 2u * path;
			if (need == 2) {
				set_ray(kid_rays[at + 1], c1_o, c1_d, RTMI_INFINITY);
				kid_nodes[at + 1].rootslot = rootslot; kid_nodes[at + 1].path = 2u * path + 1u;
			}
		}
		if (in) {
			Node &nd = nodes[k];
			st(nd.c, ndc); st(nd.coef, ndcoef);
			nd.beer = ndbeer; nd.kind = ndkind; nd.child = need ? at : kNone; nd.rootslot = rootslot; nd.path = path;
			if (has_shadow) set_ray(shadow[k], sh_o, sh_d, RTMI_INFINITY);
			else inactive(shadow[k]);
			wants_ao[k] = ao;
		}
	}
}

// The AO list: nodes whose AO answer will be used (k_shade_pre's flag, and not shadowed).  ao_at[k] = position or kNone.
// A listed node gets its AO frame at that position right away: hit point, hemisphere frame (shader.cpp:153-171) and the
// node's sample stream (mix64(seed ^ path), root ray index), 112 B.
struct __attribute__((aligned(16))) AoFrame {
	double P[3], X[3], Y[3], Z[3];
	uint64_t stream, pad;
};

__global__ void __launch_bounds__(kReserveBlock) k_ao_list(ShadeDev S, const rtmi_hit *__restrict__ hits,
                                                          const Node *__restrict__ nodes, const uint64_t *__restrict__ root, uint64_t seed,
                                                          const uint32_t *__restrict__ d_count, const uint8_t *__restrict__ wants_ao,
                                                          const uint8_t *__restrict__ shadowed, uint32_t *__restrict__ ao_at,
                                                          uint32_t *__restrict__ ao_count, AoFrame *__restrict__ frames) {
	const uint64_t count = *d_count;
	RTMI_FOR_ITEMS(k, in, count) {
		const uint32_t need = (in && wants_ao[k] && !shadowed[k]) ? 1u : 0u;
		const uint32_t at = reserve(ao_count, need);
		if (in) ao_at[k] = need ? at : kNone;
		if (need) {
			const D3 Ns = ld(S.normals + 3 * (size_t)hits[k].tri);
			const D3 P = ld(nodes[k].c); // the hit point (object.cpp:23), left there by k_shade_pre
			D3 X, Y, Z;
			ao_frame(Ns, X, Y, Z);
			AoFrame f;
			f.P[0] = P.x; f.P[1] = P.y; f.P[2] = P.z;
			f.X[0] = X.x; f.X[1] = X.y; f.X[2] = X.z;
			f.Y[0] = Y.x; f.Y[1] = Y.y; f.Y[2] = Y.z;
			f.Z[0] = Z.x; f.Z[1] = Z.y; f.Z[2] = Z.z;
			f.stream = wf_mix64(wf_mix64(seed ^ (uint64_t)nodes[k].path) ^ root[nodes[k].rootslot]);
			f.pad = 0;
			frames[at] = f;
		}
	}
}

// findOcclusionNew (shader.cpp:140-232) for the listed nodes, without the rays in memory:
// k_ao_trace       — any-hit queue over (node, sample) items; a lane builds its ray from the node's frame and the square
//                    point of (stream, sample) when it refills (:177-220), traces it with tmax = R (:226) and writes
//                    the byte at[·spp + s].  Whole waves refill: building a ray costs more than reading one.

RTMI_TRAVERSAL_KERNEL_W(4) void k_ao_trace(SceneDev scene, const AoFrame *__restrict__ frames, const uint32_t *__restrict__ ao_count,
                                           uint64_t cap_items, uint32_t spp, double R, int *__restrict__ counter, Tune tune,
                                           uint8_t *__restrict__ occ) {
	extern __shared__ int stack[];
	const uint32_t items = (uint32_t)min((unsigned long long)cap_items, (unsigned long long)*ao_count * spp);
	auto fetch = [&](uint32_t idx, RayD &Q) {
		const uint32_t at = idx / spp, smp = idx % spp;
		const AoFrame &f = frames[at];
		D3 org, dir;
		ao_ray_in_frame(ld(f.P), ld(f.X), ld(f.Y), ld(f.Z), R, wf_mix64(f.stream ^ (uint64_t)smp), org, dir);
		Q.ox = org.x; Q.oy = org.y; Q.oz = org.z;
		Q.dx = dir.x; Q.dy = dir.y; Q.dz = dir.z;
		Q.tmax = R;
		return true;
	};
	auto emit = [&](uint32_t idx, const HitQ &h, RayD &) {
		occ[idx] = h.pos != RTMI_MISS ? 1 : 0;
		return false;
	};
	trace_queue<true, false, false>(scene, stack + threadIdx.x, counter, items, tune, 0.0, 0.0, 0.0, fetch, emit);
}

// The colour of an opaque hit (shader.cpp:481-522) and its reflected child (:527-537, :643).
__global__ void __launch_bounds__(kReserveBlock) __attribute__((amdgpu_waves_per_eu(4, 4))) k_shade_post(ShadeDev S, const rtmi_ray *__restrict__ rays, const rtmi_hit *__restrict__ hits,
                             const uint32_t *__restrict__ d_count, int last, uint32_t spp, const uint8_t *__restrict__ shadowed,
                             const uint32_t *__restrict__ ao_at, const uint8_t *__restrict__ ao_occ, Node *__restrict__ nodes, uint32_t *__restrict__ kid_count,
                             rtmi_ray *__restrict__ kid_rays, Node *__restrict__ kid_nodes) {
	const uint64_t count = *d_count;
	RTMI_FOR_ITEMS(k, in, count) {
		uint32_t need = 0, ndkind = kFinal;
		D3 c0_o = d3(0, 0, 0), c0_d = d3(0, 0, 0), ndc = d3(0, 0, 0), ndcoef = d3(0, 0, 0);
		const bool mine = in && nodes[k].kind == kOpaque;
		if (mine) {
			const rtmi_ray &r = rays[k];
			const rtmi_hit h = hits[k];
			const rtmi_material &m = S.mats[h.object];
			const D3 diffuse = ld(m.diffuse), specular = ld(m.specular), ambient = ld(m.ambient), reflective = ld(m.reflectivity);
			const D3 Ns = ld(S.normals + 3 * (size_t)h.tri);
			const D3 P = ld(nodes[k].c);    // the hit point, left in the colour slot by k_shade_pre
			const D3 N = unit(Ns);          // :416
			const D3 E = unit(ld(r.o) - P); // :417
			D3 color = d3(0, 0, 0);
			if (shadowed[k]) {
				color = 0.7 * ambient; // :131-135
			} else {
				double occlusion = 1.0;
				const uint32_t at = spp ? ao_at[k] : kNone;
				if (at != kNone) {
					int c = 0;
					for (uint32_t s = 0; s < spp; ++s) c += ao_occ[(uint64_t)at * spp + s];
					occlusion = (1.0 - (c / (double)spp)); // :231
				}
				for (uint32_t i = 0; i < S.n_lights; ++i) { // :485-519
					const D3 emission = ld(S.lights[i].emission);
					const D3 L = unit(ld(S.lights[i].center) - P);
					color = color + cmul(emission, ((.85 * occlusion) * ambient) + (max2(0, dot(N, L)) * diffuse)); // :494
					if (m.roughness != 0.0) { // :496-508
						const D3 H = unit(L + E);
						const double LH = dot(L, H), NH = dot(N, H);
						const double g = sqrt((30.0 * 3016.0) - 1.0 + (LH * LH));
						double Fresnel = 0.5 * pow_ref(g - LH, 2.0) / pow_ref(g + LH, 2.0);
						Fresnel *= pow_ref(LH * (g + LH) - 1.0, 2.0) / pow_ref(LH * (g - LH) + 1.0, 2.0) + 1.0;
						const double mm = 0.4;
						double D = 1.0 / (4.0 * mm * mm * pow_ref(NH, 4.0));
						const double DPow = (pow_ref(NH, 2.0) - 1.0) / ((mm * mm) * pow_ref(NH, 2.0));
						D = pow_ref(D, DPow);
						const double G = min3(1.0, (2.0 * NH * NH) / LH, (2.0 * NH * dot(N, L)) / LH);
						color = color + (((Fresnel * D * G) / (kPiRef * dot(N, E) * dot(N, L))) * (.7 * specular));
					} else { // :510-518
						const D3 RR = (-1.0 * L) + ((2.0 * dot(L, N)) * N);
						color = color + cmul(emission, pow_ref(max2(0.0, dot(E, RR)), m.phong_exp) * specular);
					}
				}
			}
			const D3 R = (-1.0 * E) + ((2.0 * dot(E, Ns)) * Ns); // :532
			if (last) {
				ndc = color + cmul(reflective, termination()); // :643 with scene.cpp:134-137
			} else if (!nonzero(m.reflectivity)) {
				ndc = color; // colour + 0·Trace(...)
			} else {
				ndkind = kReflect;
				need = 1;
				ndc = color;
				ndcoef = reflective;
				c0_o = P + (kEpsilon * R); c0_d = R; // :534-535
			}
		}
		const uint32_t at = reserve(kid_count, need);
		if (need) {
			set_ray(kid_rays[at], c0_o, c0_d, RTMI_INFINITY);
			kid_nodes[at].rootslot = nodes[k].rootslot; kid_nodes[at].path = 2u * nodes[k].path;
		}
		if (mine) {
			Node &nd = nodes[k];
			st(nd.c, ndc);
			if (need) { st(nd.coef, ndcoef); nd.child = at; }
			nd.kind = ndkind;
		}
	}
}

// Returns of Shade, deepest generation first: shader.cpp:614, :640, :643.
__global__ void __launch_bounds__(256) k_fold(Node *__restrict__ nodes, const Node *__restrict__ kids, const uint32_t *__restrict__ d_count) {
	const uint64_t count = *d_count;
	RTMI_FOR_ITEMS(k, in, count) {
		if (!in) continue;
		const uint32_t kind = nodes[k].kind;
		if (kind < kReflect) continue;
		const uint32_t at = nodes[k].child;
		const D3 a = ld(kids[at].c), coef = ld(nodes[k].coef);
		D3 c;
		if (kind == kReflect) c = ld(nodes[k].c) + cmul(coef, a);
		else if (kind == kTir) c = cmul(coef, a);
		else {
			const double beerR = nodes[k].beer;
			c = cmul(coef, (beerR * a) + ((1 - beerR) * ld(kids[at + 1].c)));
		}
		st(nodes[k].c, c);
		nodes[k].kind = kFinal;
	}
}

// ToneMap, rasterizer.cpp:79-88
__device__ __forceinline__ uint8_t tone(double c) {
	const int v = (int)floor(256 * c);
	return (uint8_t)(v >= 255 ? 255 : v);
}
__global__ void __launch_bounds__(256) k_emit(const Node *__restrict__ nodes, const uint64_t *__restrict__ root, uint64_t n, uint64_t sub,
                       double *__restrict__ rgb, uint8_t *__restrict__ pixels) {
	RTMI_FOR_ITEMS(k, in, n) {
		if (!in) continue;
		const uint64_t dst = root[k];
		if (dst == ~0ull || nodes[k].kind == kDead) continue;
		const uint64_t at = dst - sub;
		const double r = nodes[k].c[0], g = nodes[k].c[1], b = nodes[k].c[2];
		if (rgb) { rgb[3 * at] = r; rgb[3 * at + 1] = g; rgb[3 * at + 2] = b; }
		if (pixels) { pixels[3 * at] = tone(r); pixels[3 * at + 1] = tone(g); pixels[3 * at + 2] = tone(b); }
	}
}
// A pixel's S samples are S consecutive slots: contributedColor += Trace(...) in sample order, then / S
// (rasterizer.cpp:337,343 and :394,398), then ToneMap.
__global__ void __launch_bounds__(256) k_emit_mean(const Node *__restrict__ nodes, const uint64_t *__restrict__ root, uint64_t n_pixels,
                                                   uint32_t S, double *__restrict__ rgb, uint8_t *__restrict__ pixels) {
	RTMI_FOR_ITEMS(p, in, n_pixels) {
		if (!in) continue;
		const uint64_t dst = root[p * S];
		if (dst == ~0ull) continue;
		D3 c = d3(0.0, 0.0, 0.0);
		for (uint32_t s = 0; s < S; ++s) c = c + ld(nodes[p * S + s].c);
		c = c / (double)S;
		const uint64_t at = dst / S;
		if (rgb) { rgb[3 * at] = c.x; rgb[3 * at + 1] = c.y; rgb[3 * at + 2] = c.z; }
		if (pixels) { pixels[3 * at] = tone(c.x); pixels[3 * at + 1] = tone(c.y); pixels[3 * at + 2] = tone(c.z); }
	}
}
__global__ void k_fill_colour(uint64_t n, double r, double g, double b, double *__restrict__ rgb) {
	RTMI_FOR_ITEMS(k, in, n) {
		if (in) { rgb[3 * k] = r; rgb[3 * k + 1] = g; rgb[3 * k + 2] = b; }
	}
}
__global__ void k_tone_map(const double *__restrict__ rgb, uint64_t n3, uint8_t *__restrict__ pixels) {
	RTMI_FOR_ITEMS(k, in, n3) {
		if (in) pixels[k] = tone(rgb[k]);
	}
}

// fixed grid for the striding elementwise kernels: enough waves to fill the device, never more than the items need
static inline dim3 grid_for(const rtmi_shader *sh, uint64_t cap, uint32_t block = 256) {
	const uint64_t full = (uint64_t)sh->scene->n_cus * (2048u / block);
	return dim3((uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(full, (cap + block - 1) / block)));
}

// One chunk of generation-1 rays through all generations.
struct Chunk {
	uint64_t cap; // generation-1 rays per chunk
	uint32_t levels, branch, spp;
	std::vector<uint64_t> slots; // capacity per generation
	std::vector<rtmi_ray *> rays;
	std::vector<Node *> nodes;
	rtmi_hit *hits;
	rtmi_ray *shadow;
	AoFrame *ao;
	uint8_t *shadowed, *ao_occ, *wants_ao;
	uint32_t *ao_at;
	uint64_t *root;
	uint32_t *counts; // [0, levels]: nodes per generation (one spare), [levels+1, 2·levels]: AO-list length per generation
};

static size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

// Lays the chunk's buffers out in the shader's workspace, growing it if needed.  want = generation-1 rays the caller has.
static int plan_chunk(rtmi_shader *sh, uint64_t want, uint32_t levels, uint32_t spp, Chunk &c, const char *who, uint32_t group = 1) {
	c.levels = levels;
	c.branch = sh->any_translucent ? 2u : 1u;
	c.spp = (sh->any_ambient ? spp : 0u);
	// bytes per generation-1 ray at capacity
	double per = 8.0, s = 1.0, smax = 1.0;
	for (uint32_t g = 0; g < levels; ++g) { per += s * (sizeof(rtmi_ray) + sizeof(Node)); smax = s; s *= c.branch; }
	per += smax * (sizeof(rtmi_hit) + sizeof(rtmi_ray) + 2 + 8 + (c.spp ? sizeof(AoFrame) + (double)c.spp : 0.0));
	const double budget = (double)env_uint("RTMI_SHADE_WS_MB", 8192) * 1048576.0;
	uint64_t cap = (uint64_t)(budget / per);
	cap = std::max<uint64_t>(256, cap & ~(uint64_t)255);
	cap = std::min<uint64_t>(cap, (want + 255) & ~(uint64_t)255);
	if (smax * (double)cap * std::max(1u, c.spp) > 4.0e9) cap = std::max<uint64_t>(256, (uint64_t)(4.0e9 / (smax * std::max(1u, c.spp))) & ~(uint64_t)255);
	if (smax * (double)cap * std::max(1u, c.spp) > 4.0e9) return fail(std::string(who) + ": ray tree too wide for one launch; lower max_tree_depth");
	cap = std::max<uint64_t>(group, cap / group * group); // a pixel's samples stay in one chunk
	c.cap = cap;
	c.slots.assign(levels, 0);
	size_t off = 0;
	auto take = [&](size_t bytes) { size_t at = off; off += align_up(bytes); return at; };
	std::vector<size_t> o_rays(levels), o_nodes(levels);
	uint64_t sl = cap;
	for (uint32_t g = 0; g < levels; ++g) {
		c.slots[g] = sl;
		o_rays[g] = take(sl * sizeof(rtmi_ray));
		o_nodes[g] = take(sl * sizeof(Node));
		sl *= c.branch;
	}
	const uint64_t mx = c.slots[levels - 1];
	const size_t o_hits = take(mx * sizeof(rtmi_hit)), o_sh = take(mx * sizeof(rtmi_ray)), o_shb = take(mx);
	const size_t o_ao = take(c.spp ? mx * sizeof(AoFrame) : 0), o_aob = take(c.spp ? mx * c.spp : 0);
	const size_t o_aoat = take(c.spp ? mx * sizeof(uint32_t) : 0), o_want = take(mx);
	const size_t o_root = take(cap * sizeof(uint64_t));
	const size_t o_counts = take((2 * levels + 2) * sizeof(uint32_t));
	if (off > sh->ws_bytes) {
		if (sh->ws) { (void)hipDeviceSynchronize(); (void)hipFree(sh->ws); sh->ws = nullptr; sh->ws_bytes = 0; }
		if (hipMalloc(&sh->ws, off) != hipSuccess) return fail(std::string(who) + ": hipMalloc of the workspace failed");
		sh->ws_bytes = off;
	}
	c.rays.resize(levels); c.nodes.resize(levels);
	for (uint32_t g = 0; g < levels; ++g) { c.rays[g] = (rtmi_ray *)(sh->ws + o_rays[g]); c.nodes[g] = (Node *)(sh->ws + o_nodes[g]); }
	c.hits = (rtmi_hit *)(sh->ws + o_hits);
	c.shadow = (rtmi_ray *)(sh->ws + o_sh);
	c.shadowed = (uint8_t *)(sh->ws + o_shb);
	c.ao = (AoFrame *)(sh->ws + o_ao);
	c.ao_occ = (uint8_t *)(sh->ws + o_aob);
	c.ao_at = (uint32_t *)(sh->ws + o_aoat);
	c.wants_ao = (uint8_t *)(sh->ws + o_want);
	c.root = (uint64_t *)(sh->ws + o_root);
	c.counts = (uint32_t *)(sh->ws + o_counts);
	return 0;
}

// all generations of n generation-1 rays; c.rays[0][0..n) and c.root[0..n) are filled.
static int run_chunk(rtmi_shader *sh, const Chunk &c, uint64_t n, const rtmi_trace_params &p, hipStream_t st, const char *who) {
	const ShadeDev S = {sh->scene->d_v1_orig, sh->d_normals, sh->d_mats, sh->d_lights, sh->n_lights};
	const uint32_t n_counters = 2 * c.levels + 2;
	hipLaunchKernelGGL(k_set_counts, dim3(1), dim3(64), 0, st, c.counts, n_counters, (uint32_t)n);
	uint64_t cap = n; // generation capacity for THIS chunk's n (<= c.slots[g])
	for (uint32_t g = 0; g < c.levels; ++g) {
		const int last = (g + 1 == c.levels);
		uint32_t *cnt = c.counts + g, *kid_cnt = c.counts + g + 1, *ao_cnt = c.counts + c.levels + 1 + g;
		rtmi_ray *kid_rays = last ? nullptr : c.rays[g + 1];
		Node *kid_nodes = last ? nullptr : c.nodes[g + 1];
		if (closest_hit_counted(sh->scene, c.rays[g], cap, cnt, 1, c.hits, st, who)) return -1;
		const int rad = (p.flags & RTMI_SHADE_RADIOSITY_BRANCH) ? 1 : 0;
		hipLaunchKernelGGL(k_shade_pre, grid_for(sh, cap, kReserveBlock), dim3(kReserveBlock), 0, st, S, c.rays[g], c.hits, cnt, g == 0 ? 1 : 0, last,
		                   rad, c.nodes[g], c.shadow, c.wants_ao, kid_cnt, kid_rays, kid_nodes);
		if (rad) break; // every node is final: no shadow, occlusion, children (the callers planned one level)
		if (any_hit_counted(sh->scene, c.shadow, cap, cnt, 1, c.shadowed, st, who)) return -1;
		if (c.spp) {
			hipLaunchKernelGGL(k_ao_list, grid_for(sh, cap, kReserveBlock), dim3(kReserveBlock), 0, st, S, c.hits, c.nodes[g], c.root, p.seed, cnt,
			                   c.wants_ao, c.shadowed, c.ao_at, ao_cnt, c.ao);
			int *slot = queue_slot(sh->scene, st, who);
			if (!slot) return -1;
			const uint32_t ao_grid = persistent_grid_of(sh->scene, reinterpret_cast<const void *>(k_ao_trace), cap * c.spp);
			Tune tune = launch_tune(refill_threshold(), ao_grid, cap * c.spp);
			static const unsigned refill_ao = std::min(64u, env_uint("RTMI_REFILL_AO", 64));
			tune.thresh = (int)refill_ao;
			hipLaunchKernelGGL(k_ao_trace, dim3(ao_grid), dim3(kBlock),
			                   sh->scene->stack_bytes, st, sh->scene->dev, c.ao, ao_cnt, cap * c.spp, c.spp, p.ao_radius, slot, tune, c.ao_occ);
		}
		hipLaunchKernelGGL(k_shade_post, grid_for(sh, cap, kReserveBlock), dim3(kReserveBlock), 0, st, S, c.rays[g], c.hits, cnt, last, c.spp,
		                   c.shadowed, c.ao_at, c.ao_occ, c.nodes[g], kid_cnt, kid_rays, kid_nodes);
		cap *= c.branch;
	}
	for (int g = (int)c.levels - 2; g >= 0; --g) {
		uint64_t sl = n;
		for (int q = 0; q < g; ++q) sl *= c.branch;
		hipLaunchKernelGGL(k_fold, grid_for(sh, sl), dim3(256), 0, st, c.nodes[g], c.nodes[g + 1], c.counts + g);
	}
	return launch_check(who);
}

static int check_params(const rtmi_trace_params *p, const char *who) {
	if (!p) return fail(std::string(who) + ": params is NULL");
	if (p->ao_rays && !(p->ao_radius > 0.0)) return fail(std::string(who) + ": ao_radius must be positive");
	if (p->max_tree_depth > 16) return fail(std::string(who) + ": max_tree_depth > 16");
	if (p->flags & ~RTMI_SHADE_RADIOSITY_BRANCH) return fail(std::string(who) + ": unknown flags");
	return 0;
}

} // namespace rtmi

using namespace rtmi;

extern "C" {

rtmi_shader *rtmi_shader_create(rtmi_scene *s, const double *tri_normals, const rtmi_material *mats, uint32_t n_objects,
                                const rtmi_light *lights, uint32_t n_lights) {
	const char *who = "rtmi_shader_create";
	if (check_scene(s, who)) return nullptr;
	if (!tri_normals || !mats || !lights) { fail(std::string(who) + ": NULL argument"); return nullptr; }
	if (n_lights == 0) { fail(std::string(who) + ": the shader needs at least one light (shader.cpp:120 reads light 0)"); return nullptr; }
	if (n_objects < s->n_objects) { fail(std::string(who) + ": fewer materials than the scene has objects"); return nullptr; }
	rtmi_shader *sh = new rtmi_shader();
	sh->scene = s; sh->n_objects = n_objects; sh->n_lights = n_lights;
	sh->d_normals = nullptr; sh->d_mats = nullptr; sh->d_lights = nullptr; sh->ws = nullptr; sh->ws_bytes = 0;
	sh->any_translucent = sh->any_ambient = false;
	for (uint32_t k = 0; k < n_objects; ++k) {
		const rtmi_material &m = mats[k];
		if (m.translucency[0] > 0.0 && m.translucency[1] > 0.0 && m.translucency[2] > 0.0) sh->any_translucent = true;
		if (m.ambient[0] != 0.0 || m.ambient[1] != 0.0 || m.ambient[2] != 0.0) sh->any_ambient = true;
	}
	const size_t nb = std::max<size_t>(1, s->n_tris) * 3 * sizeof(double);
	bool ok = hipMalloc(&sh->d_normals, nb) == hipSuccess && hipMalloc(&sh->d_mats, n_objects * sizeof(rtmi_material)) == hipSuccess &&
	          hipMalloc(&sh->d_lights, n_lights * sizeof(rtmi_light)) == hipSuccess;
	if (ok && s->n_tris) ok = hipMemcpy(sh->d_normals, tri_normals, s->n_tris * 3 * sizeof(double), hipMemcpyHostToDevice) == hipSuccess;
	ok = ok && hipMemcpy(sh->d_mats, mats, n_objects * sizeof(rtmi_material), hipMemcpyHostToDevice) == hipSuccess;
	ok = ok && hipMemcpy(sh->d_lights, lights, n_lights * sizeof(rtmi_light), hipMemcpyHostToDevice) == hipSuccess;
	if (!ok) {
		rtmi_shader_destroy(sh);
		fail(std::string(who) + ": allocation or upload failed");
		return nullptr;
	}
	return sh;
}

void rtmi_shader_destroy(rtmi_shader *sh) {
	if (!sh) return;
	if (sh->d_normals) (void)hipFree(sh->d_normals);
	if (sh->d_mats) (void)hipFree(sh->d_mats);
	if (sh->d_lights) (void)hipFree(sh->d_lights);
	if (sh->ws) (void)hipFree(sh->ws);
	delete sh;
}

int rtmi_trace_dev(rtmi_shader *sh, const rtmi_ray *d_rays, uint64_t n, uint32_t generation, uint64_t first_index,
                   const rtmi_trace_params *params, double *d_rgb, void *stream) {
	const char *who = "rtmi_trace_dev";
	if (!sh) return fail(std::string(who) + ": shader is NULL");
	if (check_scene(sh->scene, who) || check_params(params, who)) return -1;
	if (n == 0) return 0;
	if (!d_rays || !d_rgb) return fail(std::string(who) + ": NULL argument");
	hipStream_t st = (hipStream_t)stream;
	if (generation > params->max_tree_depth) { // scene.cpp:134-137
		hipLaunchKernelGGL(k_fill_colour, grid_for(sh, n), dim3(256), 0, st, n, 0.0, 0.0, 0.0, d_rgb);
		return launch_check(who);
	}
	Chunk c;
	const bool rad = (params->flags & RTMI_SHADE_RADIOSITY_BRANCH) != 0; // that branch never recurses and shoots no secondary ray
	if (plan_chunk(sh, n, rad ? 1u : params->max_tree_depth - generation + 1, rad ? 0u : params->ao_rays, c, who)) return -1;
	for (uint64_t at = 0; at < n; at += c.cap) {
		const uint64_t m = std::min<uint64_t>(c.cap, n - at);
		if (hipMemcpyAsync(c.rays[0], d_rays + at, m * sizeof(rtmi_ray), hipMemcpyDeviceToDevice, st) != hipSuccess)
			return fail(std::string(who) + ": copy of the rays failed");
		hipLaunchKernelGGL(k_batch_roots, grid_for(sh, m), dim3(256), 0, st, first_index + at, m, c.root);
		if (run_chunk(sh, c, m, *params, st, who)) return -1;
		hipLaunchKernelGGL(k_emit, grid_for(sh, m), dim3(256), 0, st, c.nodes[0], c.root, m, first_index, d_rgb, (uint8_t *)nullptr);
	}
	return launch_check(who);
}

int rtmi_render_frame_sampled_dev(rtmi_shader *sh, const rtmi_camera *cam, const rtmi_raster *raster, uint32_t tile_w, uint32_t tile_h,
                                  uint32_t tile_first, uint32_t tile_stride, const rtmi_trace_params *params, double *d_rgb,
                                  uint8_t *d_pixels, void *stream) {
	const char *who = "rtmi_render_frame_sampled_dev";
	if (!sh) return fail(std::string(who) + ": shader is NULL");
	if (check_scene(sh->scene, who) || check_params(params, who)) return -1;
	if (!cam || !raster || (!d_rgb && !d_pixels)) return fail(std::string(who) + ": NULL argument");
	uint32_t S = 1;
	if (raster->mode == RTMI_RASTER_ANTI_ALIASING) S = 4;
	else if (raster->mode == RTMI_RASTER_DEPTH_OF_FIELD) {
		if (raster->lens_dim == 0 || raster->lens_dim > 64) return fail(std::string(who) + ": lens_dim must be 1..64");
		S = raster->lens_dim * raster->lens_dim;
	} else if (raster->mode != RTMI_RASTER_NORMAL) return fail(std::string(who) + ": unknown raster mode");
	if (cam->x_res == 0 || cam->y_res == 0) return 0;
	if (tile_w == 0 || tile_h == 0) { tile_w = cam->x_res; tile_h = cam->y_res; }
	if (tile_stride == 0) return fail(std::string(who) + ": tile_stride must be >= 1");
	if ((uint64_t)tile_w * tile_h > 0x7fffffffull) return fail(std::string(who) + ": tile too large");
	const uint32_t tiles_x = (cam->x_res + tile_w - 1) / tile_w, tiles_y = (cam->y_res + tile_h - 1) / tile_h;
	const uint64_t n_tiles = (uint64_t)tiles_x * tiles_y;
	if (tile_first >= n_tiles) return 0;
	const uint64_t mine = (n_tiles - tile_first + tile_stride - 1) / tile_stride;
	const uint64_t n = mine * tile_w * tile_h * S;
	hipStream_t st = (hipStream_t)stream;
	if (params->max_tree_depth == 0) return fail(std::string(who) + ": max_tree_depth must be >= 1 (primary rays are generation 1)");
	const FrameGrid g = {tile_w, tile_h, tiles_x, tile_first, tile_stride};
	Chunk c;
	const bool rad = (params->flags & RTMI_SHADE_RADIOSITY_BRANCH) != 0;
	if (plan_chunk(sh, n, rad ? 1u : params->max_tree_depth, rad ? 0u : params->ao_rays, c, who, S)) return -1;
	for (uint64_t at = 0; at < n; at += c.cap) {
		const uint64_t m = std::min<uint64_t>(c.cap, n - at);
		hipLaunchKernelGGL(k_frame_rays, grid_for(sh, m), dim3(256), 0, st, *cam, *raster, S, g, at, m, c.rays[0], c.root);
		if (run_chunk(sh, c, m, *params, st, who)) return -1;
		if (S == 1) hipLaunchKernelGGL(k_emit, grid_for(sh, m), dim3(256), 0, st, c.nodes[0], c.root, m, (uint64_t)0, d_rgb, d_pixels);
		else hipLaunchKernelGGL(k_emit_mean, grid_for(sh, m / S), dim3(256), 0, st, c.nodes[0], c.root, m / S, S, d_rgb, d_pixels);
	}
	return launch_check(who);
}

int rtmi_render_frame_dev(rtmi_shader *sh, const rtmi_camera *cam, double sx, double sy, uint32_t tile_w, uint32_t tile_h,
                          uint32_t tile_first, uint32_t tile_stride, const rtmi_trace_params *params, double *d_rgb,
                          uint8_t *d_pixels, void *stream) {
	rtmi_raster ras;
	std::memset(&ras, 0, sizeof ras);
	ras.mode = RTMI_RASTER_NORMAL;
	ras.sx = sx; ras.sy = sy;
	return rtmi_render_frame_sampled_dev(sh, cam, &ras, tile_w, tile_h, tile_first, tile_stride, params, d_rgb, d_pixels, stream);
}

// RasterDetails' constructor (rasterizer.cpp:40-68) for the fields the sampled rasters read, with the reference's Vec3
// operators restated on the host.
int rtmi_raster_setup(uint32_t mode, const double eye[3], const double lookat[3], const double up[3], double vpdist,
                      const double x_win[2], const double y_win[2], rtmi_raster *out) {
	if (!eye || !lookat || !up || !x_win || !y_win || !out) return fail("rtmi_raster_setup: NULL argument");
	if (mode > RTMI_RASTER_DEPTH_OF_FIELD) return fail("rtmi_raster_setup: unknown raster mode");
	struct V { double x, y, z; };
	auto sub = [](V a, V b) { return V{a.x - b.x, a.y - b.y, a.z - b.z}; };
	auto scl = [](double s, V a) { return V{s * a.x, s * a.y, s * a.z}; };
	auto dotp = [](V a, V b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); };
	auto divs = [](V a, double c) { return V{a.x / c, a.y / c, a.z / c}; };
	auto l2 = [](V a) { return a.x * a.x + a.y * a.y + a.z * a.z; };
	auto cross = [](V a, V b) { return V{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; };
	auto unitv = [&](V a) { double d = l2(a); return d > 0.0 ? divs(a, std::sqrt(d)) : V{0, 0, 0}; };
	const V E{eye[0], eye[1], eye[2]}, L{lookat[0], lookat[1], lookat[2]}, Up{up[0], up[1], up[2]};
	const V G = unitv(sub(L, E));                 // :49
	V upG = Up;                                   // vec3.h:70-75
	{
		const double x = l2(G);
		if (x > 0.0) upG = sub(Up, scl(dotp(Up, G) / x, G));
	}
	const V U = unitv(upG), R = unitv(cross(G, U)); // :50-51
	std::memset(out, 0, sizeof *out);
	out->mode = mode;
	out->lens_dim = 7;                            // DistLensDim, :61
	out->sx = out->sy = 0.5;
	const V XG = divs(scl(0.375, R), 7.0), YG = divs(scl(0.3, U), 7.0); // Lwidth, Lheight :71-72, :62-63
	out->lookat[0] = L.x; out->lookat[1] = L.y; out->lookat[2] = L.z;
	out->R[0] = R.x; out->R[1] = R.y; out->R[2] = R.z;
	out->U[0] = U.x; out->U[1] = U.y; out->U[2] = U.z;
	out->x_gradation[0] = XG.x; out->x_gradation[1] = XG.y; out->x_gradation[2] = XG.z;
	out->y_gradation[0] = YG.x; out->y_gradation[1] = YG.y; out->y_gradation[2] = YG.z;
	out->vpdist = vpdist;
	out->xmin = x_win[0];                         // :42
	out->ymax = y_win[1];                         // :43
	return 0;
}

int rtmi_tone_map_dev(const double *d_rgb, uint64_t n, uint8_t *d_pixels, void *stream) {
	const char *who = "rtmi_tone_map_dev";
	if (n == 0) return 0;
	if (!d_rgb || !d_pixels) return fail(std::string(who) + ": NULL argument");
	hipLaunchKernelGGL(k_tone_map, dim3((uint32_t)std::min<uint64_t>(2048, (3 * n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_rgb, 3 * n, d_pixels);
	return launch_check(who);
}

} // extern "C"
