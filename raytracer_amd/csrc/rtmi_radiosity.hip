// librtmi — radiosity: visibility, form factors and the Jacobi solve (include/rtmi.h), gfx950 only.
//
// Reference: Render2/radiosity.cpp.  Form factors are evaluated in double exactly as the reference
// writes them (:305-333, :264-303) on sample points from a counter-based generator, then rounded
// once to fp32 for storage; the solve streams that fp32 matrix from HBM once per sweep.
#include "rtmi_rad_device.hpp"

#include <algorithm>

#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace rtmi {

RTMI_TRAVERSAL_KERNEL void k_visible_points(SceneDev S, PatchesDev P, uint64_t m,
                                                           const int32_t *__restrict__ ej, const double *__restrict__ xi,
                                                           const double *__restrict__ xj, uint8_t *__restrict__ vis) {
	extern __shared__ int stack[]; // rtmi_scene::stack_bytes_binary: (BVH depth + 1) entries per lane
	const uint64_t k = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
	if (k >= m) return;
	const D3 a = d3(xi[3 * k], xi[3 * k + 1], xi[3 * k + 2]), b = d3(xj[3 * k], xj[3 * k + 1], xj[3 * k + 2]);
	vis[k] = visible(S, P, (uint32_t)ej[k], a, b, stack + threadIdx.x) ? 1 : 0;
}

// ---- Jacobi sweep: HBM-bound streaming of F -------------------------------------------------------------
// A workgroup (4 waves) owns kRows consecutive rows.  The 4 waves interleave 256-column chunks of those
// rows: in every step a lane loads one float4 of each of the kRows rows (kRows independent 16-B loads
// in flight, 1 KiB per wave-instruction, rows streamed sequentially so HBM pages stay open) and the 12
// matching floats of B (3 × float4, L1/L2 hits shared by the block's rows).  Accumulation in double.
#ifndef RTMI_JROWS
#define RTMI_JROWS 8
#endif
#ifndef RTMI_JNT
#define RTMI_JNT 1
#endif
#ifndef RTMI_JBLOCK
#define RTMI_JBLOCK 256
#endif
constexpr int kRows = RTMI_JROWS;
constexpr int kJBlock = RTMI_JBLOCK, kJWaves = RTMI_JBLOCK / 64;
static_assert(kRows * 3 <= 32, "the final reduction runs over the first 32 lanes of wave 0");
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 load_F4(const float *p) {
#if RTMI_JNT
	const v4f x = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p)); // F is streamed once per sweep
	return make_float4(x.x, x.y, x.z, x.w);
#else
	return *reinterpret_cast<const float4 *>(p);
#endif
}

__global__ __launch_bounds__(kJBlock) void k_jacobi_sweep(const float *__restrict__ F, uint64_t ld, uint32_t n, uint32_t row0,
                                                         uint32_t n_rows, double rho0, double rho1, double rho2,
                                                         const float *__restrict__ E, const float *__restrict__ Bold,
                                                         float *Bnew, double *__restrict__ norms, uint32_t slots, double ca, double cb) {
	__shared__ double red[kJWaves][kRows][3];
	const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const uint32_t r0 = blockIdx.x * kRows; // local row
	double acc[kRows][3];
#pragma unroll
	for (int r = 0; r < kRows; ++r) acc[r][0] = acc[r][1] = acc[r][2] = 0.0;
	const float *rowp[kRows];
#pragma unroll
	for (int r = 0; r < kRows; ++r) rowp[r] = F + (uint64_t)min(r0 + r, n_rows - 1) * ld; // tail rows re-read the last row
	const uint32_t n4 = n & ~3u;
	for (uint32_t c = (wave * 64 + lane) * 4; c < n4; c += 4 * kJBlock) {
		float4 f[kRows];
#pragma unroll
		for (int r = 0; r < kRows; ++r) f[r] = load_F4(rowp[r] + c);
		const float4 b0 = *reinterpret_cast<const float4 *>(Bold + 3 * (size_t)c);
		const float4 b1 = *reinterpret_cast<const float4 *>(Bold + 3 * (size_t)c + 4);
		const float4 b2 = *reinterpret_cast<const float4 *>(Bold + 3 * (size_t)c + 8);
		// [c][0..2], [c+1][0..2], ... interleaved RGB
		const double B00 = b0.x, B01 = b0.y, B02 = b0.z, B10 = b0.w, B11 = b1.x, B12 = b1.y;
		const double B20 = b1.z, B21 = b1.w, B22 = b2.x, B30 = b2.y, B31 = b2.z, B32 = b2.w;
#pragma unroll
		for (int r = 0; r < kRows; ++r) {
			const double fx = f[r].x, fy = f[r].y, fz = f[r].z, fw = f[r].w;
			acc[r][0] = fma(fx, B00, acc[r][0]); acc[r][1] = fma(fx, B01, acc[r][1]); acc[r][2] = fma(fx, B02, acc[r][2]);
			acc[r][0] = fma(fy, B10, acc[r][0]); acc[r][1] = fma(fy, B11, acc[r][1]); acc[r][2] = fma(fy, B12, acc[r][2]);
			acc[r][0] = fma(fz, B20, acc[r][0]); acc[r][1] = fma(fz, B21, acc[r][1]); acc[r][2] = fma(fz, B22, acc[r][2]);
			acc[r][0] = fma(fw, B30, acc[r][0]); acc[r][1] = fma(fw, B31, acc[r][1]); acc[r][2] = fma(fw, B32, acc[r][2]);
		}
	}
	// columns n4..n-1 (n not a multiple of 4): one lane each
	if (wave == 0 && lane < (n - n4)) {
		const uint32_t c = n4 + lane;
#pragma unroll
		for (int r = 0; r < kRows; ++r) {
			const double f = rowp[r][c];
			acc[r][0] = fma(f, (double)Bold[3 * (size_t)c], acc[r][0]);
			acc[r][1] = fma(f, (double)Bold[3 * (size_t)c + 1], acc[r][1]);
			acc[r][2] = fma(f, (double)Bold[3 * (size_t)c + 2], acc[r][2]);
		}
	}
	// wave reduction (butterfly over 64 lanes), then across the 4 waves through LDS
#pragma unroll
	for (int r = 0; r < kRows; ++r)
#pragma unroll
		for (int ch = 0; ch < 3; ++ch) {
			double x = acc[r][ch];
#pragma unroll
			for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
			if (lane == 0) red[wave][r][ch] = x;
		}
	__syncthreads();
	if (wave == 0) { // whole wave takes part in the shuffles; lanes >= 24 contribute zeros
		const uint32_t r = lane / 3, ch = lane % 3;
		double d2 = 0.0, b2 = 0.0, dmax = 0.0;
		if (lane < kRows * 3 && r0 + r < n_rows) {
			double s = red[0][r][ch];
#pragma unroll
			for (int wv = 1; wv < kJWaves; ++wv) s += red[wv][r][ch];
			const size_t gi = 3 * (size_t)(row0 + r0 + r) + ch;
			const double rho = ch == 0 ? rho0 : (ch == 1 ? rho1 : rho2);
			const double e = (double)E[gi];
			// (ca, cb) = (1, 0): the Jacobi update.  Otherwise the Chebyshev step of solve_loop:
			// B_new = B_old + ca·(J(B_old) − B_old) + cb·(B_old − B_prev), where B_prev is what the ping-pong partner being
			// written still holds (the iterate before B_old).
			const double jac = e + rho * s;
			const double bo = (double)Bold[gi];
			const float bn = (ca == 1.0 && cb == 0.0) ? (float)jac : (float)(bo + ca * (jac - bo) + cb * (bo - (double)Bnew[gi]));
			const double d = (double)bn - bo;
			Bnew[gi] = bn;
			d2 = d * d;
			b2 = ((double)bn - e) * ((double)bn - e);
			dmax = fabs(d);
		}
#pragma unroll
		for (int off = 16; off > 0; off >>= 1) {
			d2 += __shfl_xor(d2, off, 64);
			b2 += __shfl_xor(b2, off, 64);
			dmax = fmax(dmax, __shfl_xor(dmax, off, 64));
		}
		if (lane == 0) { // one atomic per block and quantity; atomics on one address serialise in L2, hence the slots
			double *nm = norms + 3 * (blockIdx.x % slots);
			atomicAdd(&nm[0], d2);
			atomicAdd(&nm[1], b2);
			atomicMax(reinterpret_cast<unsigned long long *>(&nm[2]), (unsigned long long)__double_as_longlong(dmax));
		}
	}
}

// ---------------------------------------------------------------- radiosity display path
// Radiosity_Helper::trace_ray (radiosity_helper.cpp:318-365) + the barycentric half of returnQuadTreeColor
// (radiosity_helper.h:235-267): one ray per thread against EVERY element (the reference's loop; elements are staged
// through LDS in tiles, all lanes read the same record).  An element is tested on the whole line (t of either sign), inside
// ⇔ u ≥ 0 ∧ v ≥ 0 ∧ u + v ≤ 1 with (u, v) = rows of the pseudo-inverse · (P − v1) (util.cpp:126-147); the nearest |P − o|
// wins, first on ties, and a distance of exactly 0 counts as "nothing yet" (:342).  No element: Green; back face: Black.
constexpr uint32_t kDispTile = 128, kDispRec = 16;
struct DispRay { double ox, oy, oz, dx, dy, dz; bool active; };

template <class Fetch, class Emit>
__device__ __forceinline__ void display_trace(const rtmi_patches_dev_view P, const double *__restrict__ vertex_rad, uint64_t n_items,
                                              Fetch fetch, Emit emit) {
	__shared__ double tile[kDispTile * kDispRec];
	const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	DispRay r;
	r.active = k < n_items && fetch(k, r);
	const D3 O = d3(r.ox, r.oy, r.oz), D = d3(r.dx, r.dy, r.dz);
	double dist = 0.0;
	int hit = -1;
	D3 fp = d3(0, 0, 0);
	for (uint32_t base = 0; base < P.n; base += kDispTile) {
		const uint32_t cnt = min(kDispTile, P.n - base);
		__syncthreads();
		for (uint32_t w = threadIdx.x; w < cnt * kDispRec; w += blockDim.x) tile[w] = P.display[(size_t)base * kDispRec + w];
		__syncthreads();
		if (!r.active) continue;
		for (uint32_t e = 0; e < cnt; ++e) {
			const double *t = tile + e * kDispRec;
			const D3 n = d3(t[0], t[1], t[2]);
			const double denom = dot(n, D);
			if (denom != 0.0) {
				const double tt = -1.0 * ((dot(n, O) + t[12]) / denom); // :332
				const D3 Pt = (tt * D) + O;                             // :334
				double u = 0.0, v = 0.0;
				if (t[13] != 0.0) {
					const D3 q = Pt - d3(t[3], t[4], t[5]);
					u = t[6] * q.x + t[7] * q.y + t[8] * q.z;          // util.cpp:140-141
					v = t[9] * q.x + t[10] * q.y + t[11] * q.z;
				}
				if (u >= 0.0 && v >= 0.0 && u + v <= 1.0) {
					const double L = sqrt(len2(Pt - O));
					if (dist == 0.0 || dist > L) { fp = Pt; hit = (int)(base + e); dist = L; }
				}
			}
		}
	}
	if (!r.active) return;
	D3 c = d3(0.0, 1.0, 0.0); // Green
	if (hit >= 0) {
		const double *nv = P.normals + 3 * (size_t)hit;
		if (dot(d3(nv[0], nv[1], nv[2]), D) > 0.0) c = d3(0.0, 0.0, 0.0);
		else {
			const double *p = P.verts + 9 * (size_t)hit;
			const D3 a = d3(p[0], p[1], p[2]), b = d3(p[3], p[4], p[5]), cc = d3(p[6], p[7], p[8]);
			auto area = [](D3 A, D3 B, D3 C) { return sqrt(len2(cross(B - A, C - A))) / 2.0; }; // vec3.h:116-118
			const double ar = area(a, b, cc);
			const double w1 = area(fp, b, cc) / ar, w2 = area(fp, a, cc) / ar, w3 = area(fp, a, b) / ar; // radiosity_helper.h:235-238
			double f = 0.0;
			f += (w1 * vertex_rad[3 * (size_t)hit]);     // :250,256 in edge-list order: triVert1, 2, 3
			f += (w2 * vertex_rad[3 * (size_t)hit + 1]);
			f += (w3 * vertex_rad[3 * (size_t)hit + 2]);
			c = d3(f, f, f);
		}
	}
	emit(k, c);
}

// The same query through the BVH of the elements (the scene the patches were created on).  trace_ray's answer is the
// lexicographic minimum of (|P − o|, element index) over the elements whose inside test passes — unless some element is
// met at a distance of exactly 0, which the reference's loop treats as "nothing yet" (:342): such a ray (origin on an
// element's plane, inside it) falls back to the full loop.  An accepted point lies within rounding (1e-15) of its
// triangle, far inside the conservatively padded boxes, so every accepted element is among the leaves the LINE of the
// ray reaches; the boxes are tested on −tb ≤ t ≤ tb with tb the best distance so far.  Elements without a pseudo-inverse
// are inside everywhere on their plane and are tested for every ray.
__device__ __forceinline__ bool slab_line(uint32_t wx, uint32_t wy, uint32_t wz, const RayF &r, float tb) {
	const float ax = fmaf((float)(wx & 0xffffu), r.ax, r.blx), bx = fmaf((float)(wx >> 16), r.ax, r.bhx);
	const float ay = fmaf((float)(wy & 0xffffu), r.ay, r.bly), by = fmaf((float)(wy >> 16), r.ay, r.bhy);
	const float az = fmaf((float)(wz & 0xffffu), r.az, r.blz), bz = fmaf((float)(wz >> 16), r.az, r.bhz);
	const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), -tb));
	const float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), tb));
	return tn <= tf;
}

template <class Fetch, class Emit>
__device__ __forceinline__ void display_trace_bvh(const SceneDev &S, const rtmi_patches_dev_view P, const double *__restrict__ vertex_rad,
                                                  uint64_t n_items, int *stack, Fetch fetch, Emit emit) {
	const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	DispRay r;
	if (!(k < n_items && fetch(k, r))) return;
	const D3 O = d3(r.ox, r.oy, r.oz), D = d3(r.dx, r.dy, r.dz);
	double dist = 0.0;
	int hit = -1;
	bool zero = false;
	D3 fp = d3(0, 0, 0);
	auto test = [&](uint32_t e) { // the body of trace_ray's loop for element e, radiosity_helper.cpp:328-356
		const double *t = P.display + (size_t)e * kDispRec;
		const D3 n = d3(t[0], t[1], t[2]);
		const double denom = dot(n, D);
		if (denom != 0.0) {
			const double tt = -1.0 * ((dot(n, O) + t[12]) / denom);
			const D3 Pt = (tt * D) + O;
			double u = 0.0, v = 0.0;
			if (t[13] != 0.0) {
				const D3 q = Pt - d3(t[3], t[4], t[5]);
				u = t[6] * q.x + t[7] * q.y + t[8] * q.z;
				v = t[9] * q.x + t[10] * q.y + t[11] * q.z;
			}
			if (u >= 0.0 && v >= 0.0 && u + v <= 1.0) {
				const double L = sqrt(len2(Pt - O));
				if (L == 0.0) zero = true;
				if (hit < 0 || L < dist || (L == dist && (int)e < hit)) { fp = Pt; hit = (int)e; dist = L; }
			}
		}
	};
	for (uint32_t q = 0; q < P.n_noninv; ++q) test(P.noninv[q]);
	const RayF rf = make_rayf(S, O.x, O.y, O.z, D.x, D.y, D.z);
	const double dn = sqrt(len2(D));
	auto bound = [&]() { return hit < 0 ? 3.0e38f : f32_above((dist / dn) * (1.0 + 1e-6)); };
	float tb = bound();
	int sp = 0, cur = S.root_ref;
	for (uint32_t guard = 0; cur != kSentinel; ++guard) {
		if (guard >= S.guard) { give_up(S); break; } // never reached on valid input
		if (cur >= 0) {
			const QNode *np = S.nodes + cur;
			const uint4 a = np->a, b = np->b;
			const bool h0 = slab_line(a.x, a.y, a.z, rf, tb), h1 = slab_line(a.w, b.x, b.y, rf, tb);
			if (h0 && h1) { stack[sp * kBlock] = (int)b.w; ++sp; cur = (int)b.z; }
			else if (h0) cur = (int)b.z;
			else if (h1) cur = (int)b.w;
			else if (sp > 0) { --sp; cur = stack[sp * kBlock]; }
			else cur = kSentinel;
		} else {
			const uint32_t code = ~(uint32_t)cur, first = code >> 3, cnt = (code & 7u) + 1u;
			for (uint32_t i = 0; i < cnt; ++i) {
				const uint32_t e = S.tris[first + i].orig;
				if (P.display[(size_t)e * kDispRec + 13] != 0.0) test(e); // the others were tested above
			}
			tb = bound();
			if (sp > 0) { --sp; cur = stack[sp * kBlock]; }
			else cur = kSentinel;
		}
	}
	if (zero) { // the reference's order-dependent rule for a distance of exactly 0: redo its loop as written
		dist = 0.0; hit = -1;
		for (uint32_t e = 0; e < P.n; ++e) {
			const int before = hit;
			const double dbefore = dist;
			const D3 fbefore = fp;
			hit = -1;
			test(e);
			if (hit < 0) { hit = before; dist = dbefore; fp = fbefore; }             // e is not inside: keep the state
			else if (!(before < 0 || dbefore == 0.0 || dbefore > dist)) { hit = before; dist = dbefore; fp = fbefore; } // :342-351
		}
	}
	D3 c = d3(0.0, 1.0, 0.0); // Green
	if (hit >= 0) {
		const double *nv = P.normals + 3 * (size_t)hit;
		if (dot(d3(nv[0], nv[1], nv[2]), D) > 0.0) c = d3(0.0, 0.0, 0.0);
		else {
			const double *p = P.verts + 9 * (size_t)hit;
			const D3 a = d3(p[0], p[1], p[2]), b = d3(p[3], p[4], p[5]), cc = d3(p[6], p[7], p[8]);
			auto area = [](D3 A, D3 B, D3 C) { return sqrt(len2(cross(B - A, C - A))) / 2.0; };
			const double ar = area(a, b, cc);
			const double w1 = area(fp, b, cc) / ar, w2 = area(fp, a, cc) / ar, w3 = area(fp, a, b) / ar;
			double f = 0.0;
			f += (w1 * vertex_rad[3 * (size_t)hit]);
			f += (w2 * vertex_rad[3 * (size_t)hit + 1]);
			f += (w3 * vertex_rad[3 * (size_t)hit + 2]);
			c = d3(f, f, f);
		}
	}
	emit(k, c);
}

// ToneMapRadiosity, rasterizer.cpp:90-109
__device__ __forceinline__ void tone_radiosity(D3 c, uint8_t *px) {
	if (c.x == 0.0 && c.y == 1.0 && c.z == 0.0) { px[0] = 0; px[1] = 255; px[2] = 0; return; }
	const uint8_t g = (uint8_t)(int)((double)255 * (c.x / (c.x + 1.0)));
	px[0] = px[1] = px[2] = g;
}

template <bool BVH>
__global__ void __launch_bounds__(256) k_display_rays(SceneDev S, rtmi_patches_dev_view P, const double *__restrict__ vertex_rad,
                                                      const rtmi_ray *__restrict__ rays, uint64_t n, double *__restrict__ rgb) {
	extern __shared__ int disp_stack[];
	auto fetch = [&](uint64_t k, DispRay &r) {
		const rtmi_ray &q = rays[k];
		if (q.reserved != 0.0) return false;
		r.ox = q.o[0]; r.oy = q.o[1]; r.oz = q.o[2]; r.dx = q.d[0]; r.dy = q.d[1]; r.dz = q.d[2];
		return true;
	};
	auto emit = [&](uint64_t k, D3 c) { rgb[3 * k] = c.x; rgb[3 * k + 1] = c.y; rgb[3 * k + 2] = c.z; };
	if (BVH) display_trace_bvh(S, P, vertex_rad, n, disp_stack + threadIdx.x, fetch, emit);
	else display_trace(P, vertex_rad, n, fetch, emit);
}

// the pixel loop of Rasterizer::Radiosity_Raster (rasterizer.cpp:436-455): items enumerate this rank's tiles
template <bool BVH>
__global__ void __launch_bounds__(256) k_display_frame(SceneDev S, rtmi_patches_dev_view P, const double *__restrict__ vertex_rad, rtmi_camera cam,
                                                       double sx, double sy, uint32_t tile_w, uint32_t tile_h, uint32_t tiles_x,
                                                       uint32_t tile_first, uint32_t tile_stride, uint64_t n_items,
                                                       double *__restrict__ rgb, uint8_t *__restrict__ pixels) {
	auto pixel = [&](uint64_t k, uint32_t &i, uint32_t &j) {
		const uint32_t per = tile_w * tile_h, tile = tile_first + (uint32_t)(k / per) * tile_stride, in = (uint32_t)(k % per);
		j = (tile % tiles_x) * tile_w + in % tile_w;
		i = (tile / tiles_x) * tile_h + in / tile_w;
		return j < cam.x_res && i < cam.y_res;
	};
	auto fetch = [&](uint64_t k, DispRay &r) {
		uint32_t i, j;
		if (!pixel(k, i, j)) return false;
		const D3 O = d3(cam.O[0], cam.O[1], cam.O[2]), dR = d3(cam.dR[0], cam.dR[1], cam.dR[2]), dU = d3(cam.dU[0], cam.dU[1], cam.dU[2]);
		const D3 d = unit((O + (j + sx) * dR) - (i + sy) * dU); // rasterizer.cpp:452
		r.ox = cam.eye[0]; r.oy = cam.eye[1]; r.oz = cam.eye[2]; r.dx = d.x; r.dy = d.y; r.dz = d.z;
		return true;
	};
	auto emit = [&](uint64_t k, D3 c) {
		uint32_t i, j;
		pixel(k, i, j);
		const uint64_t at = (uint64_t)i * cam.x_res + j;
		if (rgb) { rgb[3 * at] = c.x; rgb[3 * at + 1] = c.y; rgb[3 * at + 2] = c.z; }
		if (pixels) tone_radiosity(c, pixels + 3 * at);
	};
	extern __shared__ int disp_stack[];
	if (BVH) display_trace_bvh(S, P, vertex_rad, n_items, disp_stack + threadIdx.x, fetch, emit);
	else display_trace(P, vertex_rad, n_items, fetch, emit);
}

__global__ void k_tone_map_radiosity(const double *__restrict__ rgb, uint64_t n, uint8_t *__restrict__ pixels) {
	const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (k < n) tone_radiosity(d3(rgb[3 * k], rgb[3 * k + 1], rgb[3 * k + 2]), pixels + 3 * k);
}

// calculatePseudoInverse (Render2/mat3x2.h:187-283) of the 3×2 matrix [m00 m01; m10 m11; m20 m21], host side, same libm
// calls: rows A, B of the pseudo-inverse; false = "not an invertable matrix" (A, B left zero)
static bool pinv3x2(double m00, double m01, double m10, double m11, double m20, double m21, double *A, double *B) {
	for (int a = 0; a < 3; ++a) A[a] = B[a] = 0.0;
	const double x = m00 * m00 + m10 * m10 + m20 * m20;
	const double z = m01 * m00 + m11 * m10 + m21 * m20;
	const double y = m00 * m01 + m10 * m11 + m20 * m21;
	const double w = m01 * m01 + m11 * m11 + m21 * m21;
	double e1 = (x + w) + std::sqrt(std::pow((x + w), 2) - (4.0 * ((x * w) - (y * z))));
	e1 /= 2.0;
	double e2 = (x + w) - std::sqrt(std::pow((x + w), 2) - (4.0 * ((x * w) - (y * z))));
	e2 /= 2.0;
	if (e1 == 0.0 || e2 == 0.0) return false;
	if (e1 == e2) {
		A[0] = m00 / e1; A[1] = m10 / e1; A[2] = m20 / e1;
		B[0] = m01 / e1; B[1] = m11 / e1; B[2] = m21 / e1;
		return true;
	}
	if (z == 0.0 && y == 0.0) {
		A[0] = m00 / e1; A[1] = m10 / e1; A[2] = m20 / e1;
		if (std::fabs((A[0] * m00 + A[1] * m10 + A[2] * m20) - 1.0) > kEpsilon) {
			A[0] = m00 / e2; A[1] = m10 / e2; A[2] = m20 / e2;
			B[0] = m01 / e1; B[1] = m11 / e1; B[2] = m21 / e1;
			return true;
		}
		B[0] = m01 / e2; B[1] = m11 / e2; B[2] = m21 / e2;
		return true;
	}
	const double l1x2 = std::fabs(x - e1), l1x1 = ((-1.0 * y) / (x - e1)) * l1x2;
	const double l2x2 = std::fabs(x - e2), l2x1 = ((-1.0 * y) / (x - e2)) * l2x2;
	const double d1 = (l1x1 * l1x1) + (l1x2 * l1x2), d2 = (l2x1 * l2x1) + (l2x2 * l2x2); // vec2.h:100-104
	const double v1x = d1 > 0.0 ? l1x1 / std::sqrt(d1) : 0, v1y = d1 > 0.0 ? l1x2 / std::sqrt(d1) : 0;
	const double v2x = d2 > 0.0 ? l2x1 / std::sqrt(d2) : 0, v2y = d2 > 0.0 ? l2x2 / std::sqrt(d2) : 0;
	double U1[3] = {m00 * v1x + m01 * v1y, m10 * v1x + m11 * v1y, m20 * v1x + m21 * v1y};
	double U2[3] = {m00 * v2x + m01 * v2y, m10 * v2x + m11 * v2y, m20 * v2x + m21 * v2y};
	auto unit3 = [](double *u) {
		const double dd = u[0] * u[0] + u[1] * u[1] + u[2] * u[2];
		if (dd > 0.0) { const double s = std::sqrt(dd); u[0] /= s; u[1] /= s; u[2] /= s; }
		else u[0] = u[1] = u[2] = 0.0;
	};
	for (int a = 0; a < 3; ++a) { U1[a] = U1[a] / std::sqrt(e1); U2[a] = U2[a] / std::sqrt(e2); }
	unit3(U1); unit3(U2);
	const double i1x = v1x * (1.0 / std::sqrt(e1)), i1y = v2x * (1.0 / std::sqrt(e2));
	const double i2x = v1y * (1.0 / std::sqrt(e1)), i2y = v2y * (1.0 / std::sqrt(e2));
	for (int a = 0; a < 3; ++a) { A[a] = i1x * U1[a] + i1y * U2[a]; B[a] = i2x * U1[a] + i2y * U2[a]; }
	return true;
}

// [v2−v1 | v3−v1] as returnUVofTriangle sets it up (util.cpp:128-134); rec = the 16-double display record of one element
static void display_record(const double *t, const double *nrm, double *rec) {
	for (int a = 0; a < 3; ++a) { rec[a] = nrm[a]; rec[3 + a] = t[a]; }
	rec[12] = ((-1.0 * nrm[0]) * t[0]) + ((-1.0 * nrm[1]) * t[1]) + ((-1.0 * nrm[2]) * t[2]); // (−1.0·n)·v1, radiosity_helper.cpp:332
	rec[14] = rec[15] = 0.0;
	// "not an invertable matrix": returnUVofTriangle then answers (0, 0)
	rec[13] = pinv3x2(t[3] - t[0], t[6] - t[0], t[4] - t[1], t[7] - t[1], t[5] - t[2], t[8] - t[2], rec + 6, rec + 9) ? 1.0 : 0.0;
}

// centerOfTriangle (Render2/radiosity_helper.h:115-142): where the medians from triVert1 and triVert2 cross, through the
// pseudo-inverse of [VLine1 | −VLine2]
static void center_of_triangle(const double *t, double *c) {
	double l1[3], l2[3];
	for (int a = 0; a < 3; ++a) {
		l1[a] = ((0.5 * (t[6 + a] - t[3 + a])) + t[3 + a]) - t[a];
		l2[a] = ((0.5 * (t[a] - t[6 + a])) + t[6 + a]) - t[3 + a];
	}
	double A[3], B[3];
	pinv3x2(l1[0], -l2[0], l1[1], -l2[1], l1[2], -l2[2], A, B);
	const double tt = A[0] * (t[3] - t[0]) + A[1] * (t[4] - t[1]) + A[2] * (t[5] - t[2]);
	for (int a = 0; a < 3; ++a) c[a] = (tt * l1[a]) + t[a];
}

// the BVH pays from a few hundred elements on; RTMI_DISPLAY_QUERY=1 forces the loop over all elements, =2 the BVH
// (tests compare them)
static bool display_uses_bvh(const rtmi_patches *P) {
	const unsigned force = env_uint("RTMI_DISPLAY_QUERY", 0);
	return force == 1 ? false : force == 2 ? true : P->n >= 256;
}

// ---- the static scratch of rtmi_radiosity_solve_noalloc_dev -----------------------------------------------------------
// Device memory that exists from the moment the code object is loaded: callers that must not allocate (HIP graph capture,
// tight serving loops) get the solver's ping-pong vector and its norm windows from here.  Every GPU the library runs on
// holds its own copy at its own address; `scratch` itself is only a host-side stand-in for the symbol, so host code uses
// the per-device pointer hipGetSymbolAddress returns — never the array's name.
constexpr size_t kScratchTmpBytes = sizeof(float) * 3 * (size_t)RTMI_NOALLOC_MAX_N;                                   // 1.5 MiB
constexpr size_t kScratchNormBytes = 3 * sizeof(double) * (size_t)kNormSlots * RTMI_NOALLOC_MAX_CHECK_EVERY;           // 96 KiB
static_assert(kScratchTmpBytes % 256 == 0, "the norms start on a 256-byte boundary");
constexpr size_t kScratchRedBytes = 256;                                                                                 // the 4 doubles a convergence check reads
static __device__ __align__(256) unsigned char scratch[kScratchTmpBytes + kScratchNormBytes + kScratchRedBytes];

constexpr int kMaxDevices = 64;
static std::mutex g_scratch_mutex;
static unsigned char *g_scratch_ptr[kMaxDevices]; // per device: the address of ITS copy of `scratch`

static unsigned char *scratch_of_current_device(const char *who) {
	int dev = -1;
	if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) { fail(std::string(who) + ": hipGetDevice failed or device index out of range"); return nullptr; }
	std::lock_guard<std::mutex> lock(g_scratch_mutex);
	if (!g_scratch_ptr[dev]) {
		void *p = nullptr;
		const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
		if (e != hipSuccess || !p) { fail(std::string(who) + ": hipGetSymbolAddress(scratch) failed: " + hipGetErrorString(e)); return nullptr; }
		g_scratch_ptr[dev] = static_cast<unsigned char *>(p);
	}
	return g_scratch_ptr[dev];
}

} // namespace rtmi

using namespace rtmi;

// One sweep; the three norms are accumulated into `slots` triples (block b into triple b % slots), which the caller adds up.
int rtmi::sweep_launch_slots(const float *d_F, uint64_t ld, uint32_t n, uint32_t row0, uint32_t n_rows, const double rho[3],
                        const float *d_E, const float *d_B_old, float *d_B_new, double *d_norms, uint32_t slots, void *stream, double ca, double cb) {
	const char *who = "rtmi_jacobi_sweep_dev";
	if (n_rows == 0) return 0;
	if (!d_F || !rho || !d_E || !d_B_old || !d_B_new || !d_norms) return fail(std::string(who) + ": NULL argument");
	if ((uint64_t)row0 + n_rows > n) return fail(std::string(who) + ": row range exceeds n");
	if (ld < n) return fail(std::string(who) + ": ld < n");
	if ((ld & 3u) || ((uintptr_t)d_F & 15u) || ((uintptr_t)d_B_old & 15u))
		return fail(std::string(who) + ": d_F / d_B_old must be 16-byte aligned and ld a multiple of 4");
	if (d_B_old == d_B_new) return fail(std::string(who) + ": Jacobi needs distinct B_old and B_new");
	const uint32_t grid = (n_rows + kRows - 1) / kRows;
	hipLaunchKernelGGL(k_jacobi_sweep, dim3(grid), dim3(kJBlock), 0, (hipStream_t)stream, d_F, ld, n, row0, n_rows, rho[0], rho[1],
	                   rho[2], d_E, d_B_old, d_B_new, d_norms, slots, ca, cb);
	return launch_check(who);
}

extern "C" {

static rtmi_patches *patches_create(rtmi_scene *s, const double *verts, const double *normals);
rtmi_patches *rtmi_patches_create(rtmi_scene *s, const double *verts, const double *normals) {
	try {
		return patches_create(s, verts, normals);
	} catch (const std::exception &e) { // no exception crosses the C boundary (std::bad_alloc of the host tables)
		fail(std::string("rtmi_patches_create: ") + e.what());
		return nullptr;
	}
}
static rtmi_patches *patches_create(rtmi_scene *s, const double *verts, const double *normals) {
	if (!s || !verts || !normals) { fail("rtmi_patches_create: NULL argument"); return nullptr; }
	int dev = -1;
	if (hipGetDevice(&dev) != hipSuccess || dev != s->device) { fail("rtmi_patches_create: wrong current device"); return nullptr; }
	if (s->n_tris == 0 || s->n_tris > 0x7fffffffu) { fail("rtmi_patches_create: scene has no (or too many) triangles"); return nullptr; }
	const uint32_t n = (uint32_t)s->n_tris;
	// patch i IS triangle i of the scene: the form-factor kernel tests "triangle j" from these vertices
	if (hash_doubles(verts, 9 * (uint64_t)n) != s->xyz_hash) { fail("rtmi_patches_create: verts is not the array the scene was built from"); return nullptr; }
	// Area(v1,v2,v3) = Length((v2−v1)^(v3−v1)) / 2, Render2/vec3.h:116-118, same operation order (host IEEE double)
	std::vector<double> area(n);
	for (uint32_t i = 0; i < n; ++i) {
		const double *p = verts + 9 * (size_t)i;
		const double ax = p[3] - p[0], ay = p[4] - p[1], az = p[5] - p[2];
		const double bx = p[6] - p[0], by = p[7] - p[1], bz = p[8] - p[2];
		const double cx = ay * bz - az * by, cy = az * bx - ax * bz, cz = ax * by - ay * bx;
		area[i] = std::sqrt(cx * cx + cy * cy + cz * cz) / 2.0;
	}
	std::vector<double> disp((size_t)n * kDispRec);
	for (uint32_t i = 0; i < n; ++i) display_record(verts + 9 * (size_t)i, normals + 3 * (size_t)i, disp.data() + (size_t)i * kDispRec);
	std::vector<uint32_t> noninv;
	for (uint32_t i = 0; i < n; ++i)
		if (disp[(size_t)i * kDispRec + 13] == 0.0) noninv.push_back(i);
	rtmi_patches *P = new rtmi_patches();
	P->scene = s;
	P->n = n;
	P->verts_hash = s->xyz_hash;
	P->d_verts = P->d_normals = P->d_area = P->d_display = nullptr;
	P->d_noninv = nullptr;
	P->n_noninv = (uint32_t)noninv.size();
	bool ok = hipMalloc(&P->d_verts, sizeof(double) * 9 * (size_t)n) == hipSuccess &&
	          hipMalloc(&P->d_normals, sizeof(double) * 3 * (size_t)n) == hipSuccess &&
	          hipMalloc(&P->d_area, sizeof(double) * (size_t)n) == hipSuccess &&
	          hipMalloc(&P->d_display, sizeof(double) * kDispRec * (size_t)n) == hipSuccess;
	ok = ok && hipMemcpy(P->d_display, disp.data(), sizeof(double) * kDispRec * (size_t)n, hipMemcpyHostToDevice) == hipSuccess;
	ok = ok && hipMalloc(&P->d_noninv, sizeof(uint32_t) * std::max<size_t>(1, noninv.size())) == hipSuccess;
	if (ok && !noninv.empty()) ok = hipMemcpy(P->d_noninv, noninv.data(), sizeof(uint32_t) * noninv.size(), hipMemcpyHostToDevice) == hipSuccess;
	ok = ok && hipMemcpy(P->d_verts, verts, sizeof(double) * 9 * (size_t)n, hipMemcpyHostToDevice) == hipSuccess;
	ok = ok && hipMemcpy(P->d_normals, normals, sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice) == hipSuccess;
	ok = ok && hipMemcpy(P->d_area, area.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice) == hipSuccess;
	if (!ok) {
		rtmi_patches_destroy(P);
		fail("rtmi_patches_create: device allocation or upload failed");
		return nullptr;
	}
	return P;
}

void rtmi_patches_destroy(rtmi_patches *P) {
	if (!P) return;
	if (P->d_verts) (void)hipFree(P->d_verts);
	if (P->d_normals) (void)hipFree(P->d_normals);
	if (P->d_area) (void)hipFree(P->d_area);
	if (P->d_display) (void)hipFree(P->d_display);
	if (P->d_noninv) (void)hipFree(P->d_noninv);
	if (P->d_chain) (void)hipFree(P->d_chain);
	if (P->d_meta) (void)hipFree(P->d_meta);
	if (P->d_dfs) (void)hipFree(P->d_dfs);
	if (P->d_gconst) (void)hipFree(P->d_gconst);
	delete P;
}

int rtmi_visible_points(rtmi_patches *P, uint64_t m, const int32_t *ej, const double *xi, const double *xj, uint8_t *vis) {
	const char *who = "rtmi_visible_points";
	if (check_patches(P, who)) return -1;
	if (m == 0) return 0;
	if (!ej || !xi || !xj || !vis) return fail(std::string(who) + ": NULL buffer");
	for (uint64_t k = 0; k < m; ++k)
		if (ej[k] < 0 || (uint32_t)ej[k] >= P->n) return fail(std::string(who) + ": patch index out of range");
	int32_t *d_ej = nullptr;
	double *d_xi = nullptr, *d_xj = nullptr;
	uint8_t *d_vis = nullptr;
	bool ok = hipMalloc(&d_ej, 4 * m) == hipSuccess && hipMalloc(&d_xi, 24 * m) == hipSuccess &&
	          hipMalloc(&d_xj, 24 * m) == hipSuccess && hipMalloc(&d_vis, m) == hipSuccess;
	ok = ok && hipMemcpy(d_ej, ej, 4 * m, hipMemcpyHostToDevice) == hipSuccess &&
	     hipMemcpy(d_xi, xi, 24 * m, hipMemcpyHostToDevice) == hipSuccess &&
	     hipMemcpy(d_xj, xj, 24 * m, hipMemcpyHostToDevice) == hipSuccess;
	int rc = ok ? 0 : fail(std::string(who) + ": device allocation or upload failed");
	if (!rc) {
		hipLaunchKernelGGL(k_visible_points, dim3((uint32_t)((m + kBlock - 1) / kBlock)), dim3(kBlock), P->scene->stack_bytes_binary, nullptr,
		                   P->scene->dev, pdev(P), m, d_ej, d_xi, d_xj, d_vis);
		rc = launch_check(who);
	}
	if (!rc && hipMemcpy(vis, d_vis, m, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(std::string(who) + ": D2H copy failed");
	if (d_ej) (void)hipFree(d_ej);
	if (d_xi) (void)hipFree(d_xi);
	if (d_xj) (void)hipFree(d_xj);
	if (d_vis) (void)hipFree(d_vis);
	return rc;
}

int rtmi_jacobi_sweep_dev(const float *d_F, uint64_t ld, uint32_t n, uint32_t row0, uint32_t n_rows, const double rho[3],
                          const float *d_E, const float *d_B_old, float *d_B_new, double *d_norms, void *stream) {
	return sweep_launch_slots(d_F, ld, n, row0, n_rows, rho, d_E, d_B_old, d_B_new, d_norms, 1, stream);
}

} // extern "C"

namespace rtmi {

// the norm slots of the last sweep of a window (and of the one before it) -> out[4] = {Σd² last, Σ(B−E)² last, Σd² before,
// max|d| last}; slots are added in index order by one lane — the order the host used to add them in, the same on every path.
// `poison` != 0 (this rank failed): the maximum becomes +inf, which every rank reads after the reduction.
__global__ void k_fold_norms(const double *__restrict__ last, const double *__restrict__ before, uint32_t slots, int poison, double *__restrict__ out) {
	if (threadIdx.x != 0 || blockIdx.x != 0) return;
	double a = 0.0, b = 0.0, c = 0.0, m = 0.0;
	for (uint32_t q = 0; q < slots; ++q) {
		a += last[3 * q];
		b += last[3 * q + 1];
		m = fmax(m, last[3 * q + 2]);
		if (before) c += before[3 * q];
	}
	out[0] = a; out[1] = b; out[2] = c;
	out[3] = poison ? (double)INFINITY : m;
}
// a rank that failed marks the triple its exchange callback is about to reduce (max over ranks): +inf reaches every rank
__global__ void k_poison_triple(double *__restrict__ t) {
	if (threadIdx.x == 0 && blockIdx.x == 0) t[2] = (double)INFINITY;
}

// The loop shared by rtmi_radiosity_solve_dev (allocates its vectors per call), rtmi_radiosity_solve_noalloc_dev (brings them
// from the library's static device array) and rtmi_radiosity_solve_sharded_dev (RCCL exchange, rtmi_comm.hip).
//   * Sweeps: Jacobi, B <- E + rho∘(F·B), for the first window (at most 3 sweeps); from the first check on (RTMI_SOLVER=jacobi
//     keeps Jacobi throughout) Chebyshev semi-iteration on the same sweep kernel:
//         B_{k+1} = B_k + ca_k·(J(B_k) − B_k) + cb_k·(B_k − B_{k-1})
//     with the standard recurrence for a spectrum of rho·F inside [a, b]: theta = 1 − (a + b)/2, delta = (b − a)/2,
//     first step ca = 1/theta, cb = 0, r_0 = delta/theta; then r_k = 1/(2·theta/delta − r_{k-1}), ca = 2·r_k/delta,
//     cb = r_k·r_{k-1}.  b is the ratio of the change norms of the last two Jacobi sweeps (a power-iteration estimate of the
//     spectral radius); a = −b/4: form-factor matrices are similar to symmetric ones (reciprocity A_i·F_ij = A_j·F_ji), their
//     eigenvalues are real, and the negative ones measured on boxes of 3.8 K and 19 K patches are small — [−b/4, b] gave the
//     fewest sweeps there (15 against Jacobi's 24 to 1e-4; [−b, b]: 19, [0, b]: 19-23) and on the synthetic matrix (11
//     against 20), tools/quick_solver.py, profiles/ab_r03.md.  A wrong interval only slows the iteration (it converges for
//     any eigenvalue above a + b − 1), and if the relative change GROWS from one check to the next the loop goes back to
//     Jacobi for good.  The sweep streams F from HBM at 0.8 of the peak either way, so time to tolerance falls with the
//     sweep count.  Same fixed point as the reference's shooting iteration (Render2/radiosity.cpp:483-563).
//   * Ranks: before the first sweep the ranks agree on a status word through the exchange (a rank whose set-up failed makes
//     EVERY rank return -1 instead of leaving its peers in a collective); a rank whose sweep launch fails later still takes
//     part in the window's exchanges and poisons the norms, so all ranks fail together at the next check.
//   * tol_rel == 0: exactly max_sweeps Jacobi sweeps, no host read and no synchronisation inside the loop (the norms stay on
//     the device) — the form a HIP graph capture can record; with a tolerance the call reads norms on the host at every check
//     and is refused while the stream is capturing.
int solve_loop(const char *who, const float *d_F, uint64_t ld, uint32_t n, uint32_t n_pad, uint32_t row0, uint32_t n_rows, const double rho[3],
               const float *d_E, float *d_B, float tol_rel, uint32_t max_sweeps, uint32_t check_every, const SolveExchange *X,
               rtmi_solve_stats *stats, hipStream_t run, float *buf0, float *buf1, double *d_norms, uint32_t slots, double *d_red,
               hipEvent_t e0, hipEvent_t e1, int setup_rc) {
	const size_t vb = sizeof(float) * 3 * (size_t)n, vpad = sizeof(float) * 3 * (size_t)n_pad;
	const size_t nb = 3 * sizeof(double) * slots;
	const bool fixed = tol_rel == 0.0f; // exactly max_sweeps sweeps, nothing read back
	hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
	const bool capturing = hipStreamIsCapturing(run, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone;
	if (capturing && (!fixed || X))
		return fail(std::string(who) + ": the stream is being captured into a graph; only tol_rel = 0 (a fixed number of sweeps, no host read) without an exchange can be captured");
	const char *solver = std::getenv("RTMI_SOLVER");
	const bool want_cheb = !fixed && !(solver && std::strcmp(solver, "jacobi") == 0);
	int rc = setup_rc;
	if (!rc && !d_red && !fixed) rc = fail(std::string(who) + ": internal: no reduction buffer");
	// B_0 = E (Render2/radiosity.cpp:506-509); rows behind n (the all-gather's padding) stay zero in both vectors
	if (!rc) {
		bool ok = true;
		if (n_pad > n) ok = hipMemsetAsync(buf0, 0, vpad, run) == hipSuccess && hipMemsetAsync(buf1, 0, vpad, run) == hipSuccess;
		ok = ok && hipMemcpyAsync(buf0, d_E, vb, hipMemcpyDeviceToDevice, run) == hipSuccess;
		if (!ok) rc = fail(std::string(who) + ": initialising B failed");
	}
	double h[4] = {0.0, 0.0, 0.0, 0.0};
	// ---- the ranks agree on a status before anyone enters the loop's collectives --------------------------------------
	if (X) {
		if (!d_red || !d_norms) return fail(std::string(who) + ": could not allocate the words the ranks agree on (this rank fails alone)");
		int xrc = 0;
		if (X->triples_are_global) { // the caller's callback: its reduction of a norm triple carries the word (max over ranks)
			xrc = hipMemsetAsync(d_norms, 0, 3 * sizeof(double), run) != hipSuccess;
			if (!xrc && rc) hipLaunchKernelGGL(k_poison_triple, dim3(1), dim3(64), 0, run, d_norms);
			// (B_0 = E on every rank, so gathering it changes nothing; a rank without vectors hands over d_B)
			if (!xrc) xrc = X->gather(X->self, rc ? d_B : buf0, d_norms, run);
			if (!xrc) xrc = hipMemcpyAsync(h, d_norms, 3 * sizeof(double), hipMemcpyDeviceToHost, run) != hipSuccess || hipStreamSynchronize(run) != hipSuccess;
			h[3] = h[2];
		} else {
			xrc = hipMemsetAsync(d_red, 0, 4 * sizeof(double), run) != hipSuccess;
			if (!xrc && rc) hipLaunchKernelGGL(k_poison_triple, dim3(1), dim3(64), 0, run, d_red + 1); // d_red[3]
			if (!xrc) xrc = X->reduce(X->self, d_red, run);
			if (!xrc) xrc = hipMemcpyAsync(h, d_red, 4 * sizeof(double), hipMemcpyDeviceToHost, run) != hipSuccess || hipStreamSynchronize(run) != hipSuccess;
		}
		if (xrc) return rc ? -1 : fail(std::string(who) + ": the exchange failed while the ranks agreed on their status");
		if (rc) return -1; // (the message of this rank's own failure stands)
		if (!(h[3] == 0.0)) return fail(std::string(who) + ": another rank failed its set-up; every rank returns");
	} else if (rc) return -1;
	float *buf[2] = {buf0, buf1};
	int at = 0; // buf[at] holds the current B
	uint32_t sweeps = 0, cheb_k = 0, accelerated = 0;
	double rel = fixed ? NAN : INFINITY, mx = fixed ? NAN : INFINITY, rel_prev = INFINITY, d2_prev_check = 0.0, radius = 0.0;
	double theta = 1.0, delta = 1.0, rho_prev = 0.0; // Chebyshev recurrence: centre and half-width of the spectrum of I − rho·F
	bool cheb = false;
	int poison = 0;
	const auto w0 = std::chrono::steady_clock::now();
	if (e0) (void)hipEventRecord(e0, run);
	while (sweeps < max_sweeps) {
		// (the first window is at most three sweeps when Chebyshev steps may follow: the third sweep's change against the
		// second's is estimate enough, and every Jacobi sweep after it is one that could have been a faster step)
		const uint32_t k = std::min(want_cheb && sweeps == 0 ? std::min(check_every, 3u) : check_every, max_sweeps - sweeps);
		if (hipMemsetAsync(d_norms, 0, nb * k, run) != hipSuccess) poison = fail(std::string(who) + ": memset failed");
		for (uint32_t i = 0; i < k; ++i) {
			double ca = 1.0, cb = 0.0;
			if (cheb) { // the first step (cb = 0) starts the recurrence from the current iterate
				++cheb_k;
				if (cheb_k == 1) { ca = 1.0 / theta; rho_prev = delta / theta; }
				else {
					const double r = 1.0 / (2.0 * theta / delta - rho_prev);
					ca = 2.0 * r / delta;
					cb = r * rho_prev;
					rho_prev = r;
				}
				++accelerated;
			}
			float *Bo = buf[(at + i) & 1], *Bn = buf[(at + i + 1) & 1];
			double *triple = d_norms + 3 * slots * i;
			if (!poison && sweep_launch_slots(d_F, ld, n, row0, n_rows, rho, d_E, Bo, Bn, triple, slots, run, ca, cb)) poison = -1;
			if (X) { // a rank that failed keeps its appointments: its peers must not wait in a collective for ever
				if (poison && X->triples_are_global) hipLaunchKernelGGL(k_poison_triple, dim3(1), dim3(64), 0, run, triple);
				if (X->gather(X->self, Bn, triple, run) != 0) return fail(std::string(who) + ": the exchange failed after sweep " + std::to_string(sweeps + i + 1) + (g_err.empty() ? "" : ": " + g_err));
			} else if (poison) return -1;
		}
		at = (at + (int)k) & 1;
		sweeps += k;
		if (fixed) {
			if (poison) return -1;
			continue;
		}
		hipLaunchKernelGGL(k_fold_norms, dim3(1), dim3(64), 0, run, d_norms + 3 * slots * (k - 1), k >= 2 ? d_norms + 3 * slots * (k - 2) : (const double *)nullptr,
		                   slots, poison, d_red);
		if (X && X->reduce && X->reduce(X->self, d_red, run) != 0) return fail(std::string(who) + ": the exchange failed at the check after sweep " + std::to_string(sweeps) + (g_err.empty() ? "" : ": " + g_err));
		if (hipMemcpyAsync(h, d_red, sizeof h, hipMemcpyDeviceToHost, run) != hipSuccess || hipStreamSynchronize(run) != hipSuccess)
			return fail(std::string(who) + ": reading the norms failed: " + hipGetErrorString(hipGetLastError()));
		if (poison) return -1;
		if (!(h[3] < INFINITY)) return fail(std::string(who) + ": a rank failed during the sweeps, or B is no longer finite; every rank returns");
		rel = h[1] > 0.0 ? std::sqrt(h[0] / h[1]) : 0.0;
		mx = h[3];
		if (rel <= (double)tol_rel) break;
		const double d2_before = k >= 2 ? h[2] : d2_prev_check;
		if (cheb && rel > rel_prev) { // never seen on a form-factor matrix; an iteration matrix with far-from-real eigenvalues could do it
			cheb = false;
		} else if (want_cheb && !cheb && cheb_k == 0 && sweeps >= 3 && d2_before > 0.0 && h[0] > 0.0) {
			// (rounded up to a multiple of 2^-10: any b near the radius serves, and the sums it comes from are added in an order
			// that differs between runs and between rank counts — the rounding makes the steps, hence B, reproducible)
			radius = std::ceil(std::sqrt(h[0] / d2_before) * 1024.0) / 1024.0;
			if (radius > 0.0 && radius < 0.9999) {
				// spectrum of rho·F taken to lie in [low·b, b] (low = −1/4; RTMI_CHEB_LOW_PERCENT overrides, −100 = symmetric)
				const char *lp = std::getenv("RTMI_CHEB_LOW_PERCENT");
				const double low = lp && *lp ? std::max(-1.0, std::min(0.5, std::atof(lp) / 100.0)) : -0.25;
				cheb = true;
				theta = 1.0 - 0.5 * (low + 1.0) * radius;
				delta = 0.5 * (1.0 - low) * radius;
			}
		}
		rel_prev = rel;
		d2_prev_check = h[0];
	}
	if (e1) (void)hipEventRecord(e1, run);
	if (buf[at] != d_B && hipMemcpyAsync(d_B, buf[at], vb, hipMemcpyDeviceToDevice, run) != hipSuccess) return fail(std::string(who) + ": copy failed");
	if (!capturing && hipStreamSynchronize(run) != hipSuccess) return fail(std::string(who) + ": synchronize failed: " + hipGetErrorString(hipGetLastError()));
	if (stats) {
		float ms = 0.f;
		if (e0 && e1 && !capturing) (void)hipEventElapsedTime(&ms, e0, e1);
		else ms = (float)(std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count() * 1e3);
		stats->sweeps = sweeps;
		stats->rel_change = rel;
		stats->max_change = mx;
		stats->seconds = ms * 1e-3;
		stats->bytes_per_sweep = 4.0 * (double)n_rows * (double)n;
		stats->accelerated_sweeps = accelerated;
		stats->spectral_radius = radius;
	}
	return 0;
}

} // namespace rtmi

extern "C" {

static int solve_args(const char *who, const float *d_F, uint32_t n, uint32_t row0, uint32_t n_rows, const double rho[3], const float *d_E,
                      const float *d_B, rtmi_exchange_fn exchange) {
	if ((!d_F && n_rows) || !rho || !d_E || !d_B) return fail(std::string(who) + ": NULL argument");
	if (n == 0) return fail(std::string(who) + ": n == 0");
	if ((uint64_t)row0 + n_rows > n) return fail(std::string(who) + ": row range exceeds n");
	// Without an exchange nobody fills the rows of B this call does not own: a partial row range would iterate on garbage.
	if (!exchange && (row0 != 0 || n_rows != n)) return fail(std::string(who) + ": a partial row range needs an exchange callback");
	return 0;
}

struct CallbackExchange { rtmi_exchange_fn fn; void *user; };
static int callback_gather(void *self, float *d_B_new, double *d_triple, hipStream_t st) {
	const CallbackExchange *c = static_cast<const CallbackExchange *>(self);
	return c->fn(c->user, d_B_new, d_triple, st) != 0 ? fail("the exchange callback returned an error") : 0;
}

int rtmi_radiosity_solve_dev(const float *d_F, uint64_t ld, uint32_t n, uint32_t row0, uint32_t n_rows, const double rho[3],
                             const float *d_E, float *d_B, float tol_rel, uint32_t max_sweeps, uint32_t check_every,
                             rtmi_exchange_fn exchange, void *user, rtmi_solve_stats *stats, void *stream) {
	const char *who = "rtmi_radiosity_solve_dev";
	if (solve_args(who, d_F, n, row0, n_rows, rho, d_E, d_B, exchange)) return -1;
	if (check_every == 0) check_every = 1;
	// Norms: per sweep of a window of check_every sweeps, `slots` triples (k_jacobi_sweep spreads its atomics over them; one
	// triple when an exchange callback reduces it across the ranks after every sweep), + the 4 words a check reads.
	const uint32_t slots = exchange ? 1u : kNormSlots;
	double *d_norms = nullptr;
	float *d_tmp = nullptr;
	hipEvent_t e0 = nullptr, e1 = nullptr;
	int rc = 0;
	// (the small block first: it also carries the status word the ranks agree on, so a rank that then fails to get its vector
	// can still say so; RTMI_DEBUG_FAIL_SOLVE_SETUP=1 makes this rank fail here — tests)
	if (hipMalloc(&d_norms, 3 * sizeof(double) * slots * check_every + 4 * sizeof(double)) != hipSuccess) d_norms = nullptr;
	if (!d_norms || env_uint("RTMI_DEBUG_FAIL_SOLVE_SETUP", 0) || hipMalloc(&d_tmp, sizeof(float) * 3 * (size_t)n) != hipSuccess)
		rc = fail(std::string(who) + ": hipMalloc failed");
	if (!rc && (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)) rc = fail(std::string(who) + ": hipEventCreate failed");
	// (A hipGraph of one window was measured at 3 816 patches: 28.5 vs 28.6 µs per sweep — the launches already run back to
	// back on the stream and the sweep itself takes the time — and is not used.)
	CallbackExchange cb = {exchange, user};
	const SolveExchange X = {&cb, callback_gather, nullptr, true};
	rc = solve_loop(who, d_F, ld, n, n, row0, n_rows, rho, d_E, d_B, tol_rel, max_sweeps, check_every, exchange ? &X : nullptr, stats,
	                (hipStream_t)stream, d_B, d_tmp, d_norms, slots, d_norms ? d_norms + 3 * (size_t)slots * check_every : nullptr, e0, e1, rc);
	if (e0) (void)hipEventDestroy(e0);
	if (e1) (void)hipEventDestroy(e1);
	if (d_tmp) (void)hipFree(d_tmp);
	if (d_norms) (void)hipFree(d_norms);
	return rc;
}

int rtmi_radiosity_solve_noalloc_dev(const float *d_F, uint64_t ld, uint32_t n, const double rho[3], const float *d_E, float *d_B,
                                     float tol_rel, uint32_t max_sweeps, uint32_t check_every, rtmi_solve_stats *stats, void *stream) {
	const char *who = "rtmi_radiosity_solve_noalloc_dev";
	if (solve_args(who, d_F, n, 0, n, rho, d_E, d_B, nullptr)) return -1;
	if (n > RTMI_NOALLOC_MAX_N)
		return fail(std::string(who) + ": n = " + std::to_string(n) + " exceeds RTMI_NOALLOC_MAX_N = " + std::to_string(RTMI_NOALLOC_MAX_N) +
		            " (the static scratch array); use rtmi_radiosity_solve_dev");
	if (check_every == 0) check_every = 1;
	if (check_every > RTMI_NOALLOC_MAX_CHECK_EVERY)
		return fail(std::string(who) + ": check_every exceeds RTMI_NOALLOC_MAX_CHECK_EVERY = " + std::to_string(RTMI_NOALLOC_MAX_CHECK_EVERY));
	unsigned char *base = scratch_of_current_device(who);
	if (!base) return -1;
	float *d_tmp = reinterpret_cast<float *>(base);                       // B ping-pong partner: fully written by sweep 1 before it is read
	double *d_norms = reinterpret_cast<double *>(base + kScratchTmpBytes); // zeroed per window by solve_loop
	double *d_red = reinterpret_cast<double *>(base + kScratchTmpBytes + kScratchNormBytes); // the 4 words of a check
	return solve_loop(who, d_F, ld, n, n, 0, n, rho, d_E, d_B, tol_rel, max_sweeps, check_every, nullptr, stats, (hipStream_t)stream,
	                  d_B, d_tmp, d_norms, kNormSlots, d_red, nullptr, nullptr, 0);
}

int rtmi_radiosity_trace_dev(rtmi_patches *P, const double *d_vertex_radiosity, const rtmi_ray *d_rays, uint64_t n, double *d_rgb,
                             void *stream) {
	const char *who = "rtmi_radiosity_trace_dev";
	if (check_patches(P, who)) return -1;
	if (n == 0) return 0;
	if (!d_vertex_radiosity || !d_rays || !d_rgb) return fail(std::string(who) + ": NULL argument");
	if (n > (uint64_t)0x7fffffff * 256) return fail(std::string(who) + ": too many rays for one launch");
	const rtmi_patches_dev_view V = {P->d_verts, P->d_normals, P->d_display, P->n, P->d_noninv, P->n_noninv};
	if (display_uses_bvh(P))
		hipLaunchKernelGGL(k_display_rays<true>, dim3((uint32_t)((n + 255) / 256)), dim3(256), P->scene->stack_bytes, (hipStream_t)stream, P->scene->dev, V,
		                   d_vertex_radiosity, d_rays, n, d_rgb);
	else
		hipLaunchKernelGGL(k_display_rays<false>, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, P->scene->dev, V, d_vertex_radiosity,
		                   d_rays, n, d_rgb);
	return launch_check(who);
}

// returnQuadTreeColor's vertex values (Render2/radiosity_helper.h:156-213), see include/rtmi.h.  Host only.
int rtmi_radiosity_vertex_values(const double *verts, uint32_t n, const double *B, double *out) {
	const char *who = "rtmi_radiosity_vertex_values";
	if (n == 0) return 0;
	if (!verts || !B || !out) return fail(std::string(who) + ": NULL argument");
	for (size_t k = 0; k < 9 * (size_t)n; ++k)
		if (!std::isfinite(verts[k])) return fail(std::string(who) + ": non-finite vertex coordinate"); // (the corners are sorted by position)
	try {
		std::vector<double> centre(3 * (size_t)n);
		for (uint32_t e = 0; e < n; ++e) center_of_triangle(verts + 9 * (size_t)e, centre.data() + 3 * (size_t)e);
		// who has this vertex: corners sorted by their three doubles (−0 = +0), then by element — the order they are summed in
		struct Corner { double p[3]; uint32_t e; };
		std::vector<Corner> corners(3 * (size_t)n);
		for (uint32_t e = 0; e < n; ++e)
			for (int k = 0; k < 3; ++k) {
				Corner &c = corners[3 * (size_t)e + k];
				for (int a = 0; a < 3; ++a) c.p[a] = verts[9 * (size_t)e + 3 * k + a] + 0.0;
				c.e = e;
			}
		auto less = [](const Corner &x, const Corner &y) {
			for (int a = 0; a < 3; ++a) if (x.p[a] != y.p[a]) return x.p[a] < y.p[a];
			return x.e < y.e;
		};
		std::sort(corners.begin(), corners.end(), less);
		auto same = [](const Corner &x, const Corner &y) { return x.p[0] == y.p[0] && x.p[1] == y.p[1] && x.p[2] == y.p[2]; };
		auto length = [](const double *a, const double *b) { // Length(a − b), vec3.h:14
			const double x = a[0] - b[0], y = a[1] - b[1], z = a[2] - b[2];
			return std::sqrt(x * x + y * y + z * z);
		};
		for (size_t lo = 0; lo < corners.size();) {
			size_t hi = lo + 1;
			while (hi < corners.size() && same(corners[lo], corners[hi])) ++hi;
			for (size_t q = lo; q < hi; ++q) {
				const uint32_t e = corners[q].e;
				if (q > lo && corners[q - 1].e == e) continue; // a degenerate element with the point twice: handled with its first corner
				const double *p = corners[q].p, *ce = centre.data() + 3 * (size_t)e;
				double acc = 0.0;
				uint32_t cnt = 0;
				for (size_t r = lo; r < hi; ++r) {
					const uint32_t a = corners[r].e;
					if (a == e || (r > lo && corners[r - 1].e == a)) continue;
					const double *ca = centre.data() + 3 * (size_t)a;
					const double l1 = length(ce, p), l2 = length(p, ca), full = l1 + l2;              // :200
					acc += ((1.0 - (l1 / full)) * B[e]) + ((1.0 - (l2 / full)) * B[a]);               // :202-203
					++cnt;                                                                             // :205
				}
				const double value = cnt ? acc / cnt : B[e];                                          // :212, :255-257
				for (int k = 0; k < 3; ++k) {
					const double *v = verts + 9 * (size_t)e + 3 * k;
					if (v[0] + 0.0 == p[0] && v[1] + 0.0 == p[1] && v[2] + 0.0 == p[2]) out[3 * (size_t)e + k] = value;
				}
			}
			lo = hi;
		}
		return 0;
	} catch (const std::exception &ex) {
		return fail(std::string(who) + ": " + ex.what());
	}
}

int rtmi_radiosity_raster_dev(rtmi_patches *P, const double *d_vertex_radiosity, const rtmi_camera *cam, double sx, double sy,
                              uint32_t tile_w, uint32_t tile_h, uint32_t tile_first, uint32_t tile_stride, double *d_rgb,
                              uint8_t *d_pixels, void *stream) {
	const char *who = "rtmi_radiosity_raster_dev";
	if (check_patches(P, who)) return -1;
	if (!d_vertex_radiosity || !cam || (!d_rgb && !d_pixels)) return fail(std::string(who) + ": NULL argument");
	if (cam->x_res == 0 || cam->y_res == 0) return 0;
	if (tile_w == 0 || tile_h == 0) { tile_w = cam->x_res; tile_h = cam->y_res; }
	if (tile_stride == 0) return fail(std::string(who) + ": tile_stride must be >= 1");
	if ((uint64_t)tile_w * tile_h > 0x7fffffffull) return fail(std::string(who) + ": tile too large");
	const uint32_t tiles_x = (cam->x_res + tile_w - 1) / tile_w, tiles_y = (cam->y_res + tile_h - 1) / tile_h;
	const uint64_t n_tiles = (uint64_t)tiles_x * tiles_y;
	if (tile_first >= n_tiles) return 0;
	const uint64_t n = ((n_tiles - tile_first + tile_stride - 1) / tile_stride) * tile_w * tile_h;
	if (n > (uint64_t)0x7fffffff * 256) return fail(std::string(who) + ": frame too large for one launch");
	const rtmi_patches_dev_view V = {P->d_verts, P->d_normals, P->d_display, P->n, P->d_noninv, P->n_noninv};
	if (display_uses_bvh(P))
		hipLaunchKernelGGL(k_display_frame<true>, dim3((uint32_t)((n + 255) / 256)), dim3(256), P->scene->stack_bytes, (hipStream_t)stream, P->scene->dev, V,
		                   d_vertex_radiosity, *cam, sx, sy, tile_w, tile_h, tiles_x, tile_first, tile_stride, n, d_rgb, d_pixels);
	else
		hipLaunchKernelGGL(k_display_frame<false>, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, P->scene->dev, V, d_vertex_radiosity,
		                   *cam, sx, sy, tile_w, tile_h, tiles_x, tile_first, tile_stride, n, d_rgb, d_pixels);
	return launch_check(who);
}

int rtmi_tone_map_radiosity_dev(const double *d_rgb, uint64_t n, uint8_t *d_pixels, void *stream) {
	const char *who = "rtmi_tone_map_radiosity_dev";
	if (n == 0) return 0;
	if (!d_rgb || !d_pixels) return fail(std::string(who) + ": NULL argument");
	hipLaunchKernelGGL(k_tone_map_radiosity, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_rgb, n, d_pixels);
	return launch_check(who);
}

} // extern "C"
