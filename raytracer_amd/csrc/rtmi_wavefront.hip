// librtmi — the secondary-ray generators of the wavefront: the batch producers that sit between two ray-query
// calls so that a frame never leaves the device (primary hits -> shadow / ambient-occlusion rays -> any-hit).
// Reference: Render2/shader.cpp (generateNormalShadows :105-138, findOcclusionNew :140-232), all in double in the
// reference's operation order.  gfx950 only.
#include "rtmi_host.hpp"

namespace rtmi {

// generateNormalShadows (shader.cpp:116-127): origin P + 0.01·normal, direction Unit(C − P)
__global__ void k_shadow_rays(const double *__restrict__ v1_orig, const double *__restrict__ normals,
                              const rtmi_ray *__restrict__ rays, const rtmi_hit *__restrict__ hits, uint64_t n, double cx,
                              double cy, double cz, rtmi_ray *__restrict__ out) {
	const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= n) return;
	rtmi_ray q;
	const rtmi_hit h = hits[k];
	if (h.tri == RTMI_MISS) {
		inactive(q);
	} else {
		const D3 nrm = d3(normals[3 * (size_t)h.tri], normals[3 * (size_t)h.tri + 1], normals[3 * (size_t)h.tri + 2]);
		const D3 P = hit_point(rays[k], v1_orig + 3 * (size_t)h.tri, nrm);
		const D3 dir = unit(d3(cx, cy, cz) - P);
		const D3 org = P + (0.01 * nrm);
		q.o[0] = org.x; q.o[1] = org.y; q.o[2] = org.z;
		q.d[0] = dir.x; q.d[1] = dir.y; q.d[2] = dir.z;
		q.tmax = RTMI_INFINITY;
		q.reserved = 0.0;
	}
	out[k] = q;
}

// findOcclusionNew (shader.cpp:140-232): one lane per (hit, sample); the square point the reference draws from an
// unseeded mt19937 (:148-150,177) comes from the counter-based stream (seed, ray index, sample).
__global__ void k_ao_rays(const double *__restrict__ v1_orig, const double *__restrict__ normals,
                          const rtmi_ray *__restrict__ rays, const rtmi_hit *__restrict__ hits, uint64_t n, uint32_t spp,
                          double R, uint64_t seed, rtmi_ray *__restrict__ out) {
	const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (idx >= n * spp) return;
	const uint64_t k = idx / spp;
	const uint32_t s = (uint32_t)(idx % spp);
	rtmi_ray q;
	const rtmi_hit h = hits[k];
	if (h.tri == RTMI_MISS) {
		inactive(q);
	} else {
		const D3 nrm = d3(normals[3 * (size_t)h.tri], normals[3 * (size_t)h.tri + 1], normals[3 * (size_t)h.tri + 2]);
		const D3 P = hit_point(rays[k], v1_orig + 3 * (size_t)h.tri, nrm);
		const uint64_t hsh = wf_mix64(wf_mix64(seed ^ k) ^ (uint64_t)s);
		D3 org, dir;
		ao_ray(P, nrm, R, hsh, org, dir);
		q.o[0] = org.x; q.o[1] = org.y; q.o[2] = org.z;
		q.d[0] = dir.x; q.d[1] = dir.y; q.d[2] = dir.z;
		q.tmax = R; // occluded ⇔ Cast ∧ distance < R (:226)
		q.reserved = 0.0;
	}
	out[idx] = q;
}

// generateSoftShadows (shader.cpp:41-103; the shipped Shade does not call it, :475,521): n_points rays per hit toward a
// grid on the plane z = light centre: the reference halves n_points while even -> divisionsInX × divisionsInY points
// (100 -> 25 × 4), point (i, j) = (Cx − 1 + (2j+1)/divY, Cy − 1 + (2i+1)/divX, Cz) (:60-71); direction Unit(point − P),
// origin P + 0.05·direction (:90-91).  Ray k·n_points + i·divY + j.
__global__ void k_soft_shadow_rays(const double *__restrict__ v1_orig, const double *__restrict__ normals,
                                   const rtmi_ray *__restrict__ rays, const rtmi_hit *__restrict__ hits, uint64_t n, uint32_t n_points,
                                   uint32_t div_x, uint32_t div_y, double cx, double cy, double cz, rtmi_ray *__restrict__ out) {
	const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (idx >= n * n_points) return;
	const uint64_t k = idx / n_points;
	const uint32_t s = (uint32_t)(idx % n_points);
	const rtmi_hit h = hits[k];
	if (h.tri == RTMI_MISS) {
		inactive(out[idx]);
		return;
	}
	const D3 nrm = d3(normals[3 * (size_t)h.tri], normals[3 * (size_t)h.tri + 1], normals[3 * (size_t)h.tri + 2]);
	const D3 P = hit_point(rays[k], v1_orig + 3 * (size_t)h.tri, nrm);
	const uint32_t i = s / div_y, j = s % div_y;
	const D3 target = d3(cx - 1 + ((j * 2 + 1) / (double)div_y), cy - 1 + ((i * 2 + 1) / (double)div_x), cz); // :69
	const D3 dir = unit(target - P);
	set_ray(out[idx], P + (0.05 * dir), dir, RTMI_INFINITY);
}

// :93-100: a ray contributes iff Cast hits something whose object differs from the shaded hit's object;
// factor = 1 − contributors / n_points (the colour is multiplied by it)
__global__ void k_soft_shadow_factor(const rtmi_hit *__restrict__ hits, const rtmi_hit *__restrict__ shadow_hits, uint64_t n,
                                     uint32_t n_points, double *__restrict__ out) {
	const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= n) return;
	const uint32_t mine = hits[k].object;
	int c = 0;
	if (hits[k].tri != RTMI_MISS)
		for (uint32_t s = 0; s < n_points; ++s) {
			const rtmi_hit h = shadow_hits[k * n_points + s];
			c += (h.tri != RTMI_MISS && h.object != mine) ? 1 : 0;
		}
	out[k] = c > 0 ? (1.0 - ((double)c) / ((double)n_points)) : 1.0;
}

// 1 − occlusions / spp per hit (shader.cpp:231)
__global__ void k_occlusion_fraction(const uint8_t *__restrict__ occ, uint64_t n, uint32_t spp, double *__restrict__ out) {
	const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= n) return;
	int c = 0;
	for (uint32_t s = 0; s < spp; ++s) c += occ[k * spp + s];
	out[k] = (1.0 - (c / (double)spp));
}

} // namespace rtmi

using namespace rtmi;

extern "C" {

int rtmi_shadow_rays_dev(rtmi_scene *s, const double *d_normals, const rtmi_ray *d_rays, const rtmi_hit *d_hits, uint64_t n,
                         const double light_center[3], rtmi_ray *d_out, void *stream) {
	const char *who = "rtmi_shadow_rays_dev";
	if (check_scene(s, who)) return -1;
	if (n == 0) return 0;
	if (!d_normals || !d_rays || !d_hits || !light_center || !d_out) return fail(std::string(who) + ": NULL argument");
	if (n > (uint64_t)0x7fffffff * 256) return fail(std::string(who) + ": too many rays for one launch");
	hipLaunchKernelGGL(k_shadow_rays, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, s->d_v1_orig, d_normals,
	                   d_rays, d_hits, n, light_center[0], light_center[1], light_center[2], d_out);
	return launch_check(who);
}

int rtmi_ao_rays_dev(rtmi_scene *s, const double *d_normals, const rtmi_ray *d_rays, const rtmi_hit *d_hits, uint64_t n,
                     uint32_t spp, double radius, uint64_t seed, rtmi_ray *d_out, void *stream) {
	const char *who = "rtmi_ao_rays_dev";
	if (check_scene(s, who)) return -1;
	if (n == 0 || spp == 0) return 0;
	if (!d_normals || !d_rays || !d_hits || !d_out) return fail(std::string(who) + ": NULL argument");
	if (!(radius > 0.0)) return fail(std::string(who) + ": radius must be positive");
	const uint64_t total = n * spp;
	if (total > (uint64_t)0x7fffffff * 256) return fail(std::string(who) + ": too many rays for one launch");
	hipLaunchKernelGGL(k_ao_rays, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, s->d_v1_orig, d_normals,
	                   d_rays, d_hits, n, spp, radius, seed, d_out);
	return launch_check(who);
}

int rtmi_soft_shadow_rays_dev(rtmi_scene *s, const double *d_normals, const rtmi_ray *d_rays, const rtmi_hit *d_hits, uint64_t n,
                              const double light_center[3], uint32_t n_points, rtmi_ray *d_out, void *stream) {
	const char *who = "rtmi_soft_shadow_rays_dev";
	if (check_scene(s, who)) return -1;
	if (n == 0 || n_points == 0) return 0;
	if (!d_normals || !d_rays || !d_hits || !light_center || !d_out) return fail(std::string(who) + ": NULL argument");
	const uint64_t total = n * n_points;
	if (total > (uint64_t)0x7fffffff * 256) return fail(std::string(who) + ": too many rays for one launch");
	uint32_t div_x = n_points, div_y = 1; // shader.cpp:60-63
	while (div_x % 2 == 0) { div_x /= 2; div_y *= 2; }
	hipLaunchKernelGGL(k_soft_shadow_rays, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, s->d_v1_orig, d_normals,
	                   d_rays, d_hits, n, n_points, div_x, div_y, light_center[0], light_center[1], light_center[2], d_out);
	return launch_check(who);
}

int rtmi_soft_shadow_factor_dev(const rtmi_hit *d_hits, const rtmi_hit *d_shadow_hits, uint64_t n, uint32_t n_points, double *d_out,
                                void *stream) {
	const char *who = "rtmi_soft_shadow_factor_dev";
	if (n == 0) return 0;
	if (!d_hits || !d_shadow_hits || !d_out || n_points == 0) return fail(std::string(who) + ": NULL argument or n_points == 0");
	hipLaunchKernelGGL(k_soft_shadow_factor, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_hits, d_shadow_hits, n,
	                   n_points, d_out);
	return launch_check(who);
}

int rtmi_occlusion_fraction_dev(const uint8_t *d_occluded, uint64_t n, uint32_t spp, double *d_out, void *stream) {
	const char *who = "rtmi_occlusion_fraction_dev";
	if (n == 0) return 0;
	if (!d_occluded || !d_out || spp == 0) return fail(std::string(who) + ": NULL argument or spp == 0");
	hipLaunchKernelGGL(k_occlusion_fraction, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_occluded, n, spp,
	                   d_out);
	return launch_check(who);
}

} // extern "C"
