/* TEST INFRASTRUCTURE — not part of the product.  See rt_oracle.h.
 *
 * Plain-C restatement of the reference's hot path in IEEE double, in the
 * reference's own operation order (Vec3 operators: Render2/vec3.h).  Build with
 * -ffp-contract=off.  Parity status: PINNED against the compiled reference by
 * tests/test_oracle_golden.py (vectors from oracle/gen_golden.py).
 */
#include "rt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- vec3.h restated ------------------------------------------------------ */
typedef struct { double x, y, z; } v3;
static inline v3 ld3(const double *p) { v3 a = {p[0], p[1], p[2]}; return a; }
static inline void st3(double *p, v3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
static inline v3 add(v3 a, v3 b) { v3 r = {a.x + b.x, a.y + b.y, a.z + b.z}; return r; }          /* vec3.h:22 */
static inline v3 sub(v3 a, v3 b) { v3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }          /* vec3.h:26 */
static inline v3 scl(double s, v3 a) { v3 r = {s * a.x, s * a.y, s * a.z}; return r; }            /* vec3.h:34,38 */
static inline double dot(v3 a, v3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }          /* vec3.h:42 */
static inline v3 divs(v3 a, double c) { v3 r = {a.x / c, a.y / c, a.z / c}; return r; }           /* vec3.h:46 */
static inline v3 cross(v3 a, v3 b) {                                                              /* vec3.h:50 */
	v3 r = {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
	return r;
}
static inline double len2(v3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }                      /* vec3.h:14 */
static inline double len(v3 a) { return sqrt(len2(a)); }                                          /* vec3.h:18 */
static inline v3 unit(v3 a) {                                                                     /* vec3.h:98 */
	double d = len2(a);
	v3 z = {0, 0, 0};
	return d > 0.0 ? divs(a, sqrt(d)) : z;
}
static inline v3 remove_parallel(v3 a, v3 b) {                                                    /* vec3.h:70-75 */
	double x = len2(b);
	if (x > 0.0) return sub(a, scl(dot(a, b) / x, b));
	return a;
}

/* ---- object.cpp:29-60 ------------------------------------------------------ */
static inline double mt(v3 orig, v3 dir, v3 v0, v3 v1, v3 v2, double *u, double *v) {
	v3 e1 = sub(v1, v0);
	v3 e2 = sub(v2, v0);
	v3 pvec = cross(dir, e2);
	double det = dot(e1, pvec);
	if (det < 1e-8 && det > -1e-8) return 0.0;
	double inv_det = 1.0 / det;
	v3 tvec = sub(orig, v0);
	*u = dot(tvec, pvec) * inv_det;
	if (*u < 0.0 || *u > 1.0) return 0.0;
	v3 qvec = cross(tvec, e1);
	*v = dot(dir, qvec) * inv_det;
	if (*v < 0.0 || *u + *v > 1.0) return 0.0;
	return dot(e2, qvec) * inv_det;
}

double rto_mt(const double o[3], const double d[3], const double v0[3], const double v1[3], const double v2[3],
              double *u, double *v) {
	return mt(ld3(o), ld3(d), ld3(v0), ld3(v1), ld3(v2), u, v);
}

void rto_mt_batch(long n, const double *o, const double *d, const double *v0, const double *v1, const double *v2,
                  double *t, double *u, double *v) {
	for (long i = 0; i < n; ++i) {
		double uu = 0.0, vv = 0.0;
		t[i] = mt(ld3(o + 3 * i), ld3(d + 3 * i), ld3(v0 + 3 * i), ld3(v1 + 3 * i), ld3(v2 + 3 * i), &uu, &vv);
		u[i] = uu;
		v[i] = vv;
	}
}

/* object.cpp:143-168 semantics: keep strict minimum of positive t, first wins */
static inline int32_t closest(long n_tris, const double *tris, v3 O, v3 D, double tmax, double *bt, double *bu,
                              double *bv) {
	double best = tmax, u0 = 0, v0 = 0;
	int32_t bi = -1;
	for (long i = 0; i < n_tris; ++i) {
		const double *p = tris + 9 * i;
		double uu = 0, vv = 0;
		double t = mt(O, D, ld3(p), ld3(p + 3), ld3(p + 6), &uu, &vv);
		if (t > 0.0 && best > t) { best = t; bi = (int32_t)i; u0 = uu; v0 = vv; }
	}
	*bt = bi >= 0 ? best : 0.0;
	*bu = u0;
	*bv = v0;
	return bi;
}

void rto_closest_hit(long n_tris, const double *tris, long n_rays, const double *o, const double *d,
                     const double *tmax, int32_t *tri, double *t, double *u, double *v) {
#pragma omp parallel for schedule(dynamic, 64)
	for (long k = 0; k < n_rays; ++k)
		tri[k] = closest(n_tris, tris, ld3(o + 3 * k), ld3(d + 3 * k), tmax ? tmax[k] : RTO_INFINITY, &t[k], &u[k],
		                 &v[k]);
}

void rto_any_hit(long n_tris, const double *tris, long n_rays, const double *o, const double *d, const double *tmax,
                 uint8_t *occluded) {
#pragma omp parallel for schedule(dynamic, 64)
	for (long k = 0; k < n_rays; ++k) {
		v3 O = ld3(o + 3 * k), D = ld3(d + 3 * k);
		double lim = tmax ? tmax[k] : RTO_INFINITY;
		uint8_t occ = 0;
		for (long i = 0; i < n_tris && !occ; ++i) {
			const double *p = tris + 9 * i;
			double uu = 0, vv = 0;
			double tt = mt(O, D, ld3(p), ld3(p + 3), ld3(p + 6), &uu, &vv);
			if (tt > 0.0 && tt < lim) occ = 1;
		}
		occluded[k] = occ;
	}
}

/* ---- rasterizer.cpp:40-56, 230 --------------------------------------------- */
void rto_raster_details(const double *c, uint32_t x_res, uint32_t y_res, double *out18) {
	v3 eye = ld3(c), lookat = ld3(c + 3), up = ld3(c + 6);
	double vpdist = c[9], xmin = c[10], ymax = c[13];
	double width = c[11] - c[10];  /* interval.h:15-20 Len() */
	if (!(width > 0.0)) width = 0.0;
	double height = c[13] - c[12];
	if (!(height > 0.0)) height = 0.0;
	v3 G = unit(sub(lookat, eye));
	v3 U = unit(remove_parallel(up, G));
	v3 R = unit(cross(G, U));
	v3 O = add(add(scl(vpdist, G), scl(xmin, R)), scl(ymax, U));
	v3 dR = divs(scl(width, R), (double)x_res);
	v3 dU = divs(scl(height, U), (double)y_res);
	st3(out18, G); st3(out18 + 3, U); st3(out18 + 6, R); st3(out18 + 9, O); st3(out18 + 12, dR); st3(out18 + 15, dU);
}

void rto_primary_dirs(const double *rd, long n, const uint32_t *pix, double *dir) {
	v3 O = ld3(rd + 9), dR = ld3(rd + 12), dU = ld3(rd + 15);
	for (long k = 0; k < n; ++k) {
		uint32_t i = pix[2 * k], j = pix[2 * k + 1];
		st3(dir + 3 * k, unit(sub(add(O, scl(j + 0.5, dR)), scl(i + 0.5, dU))));
	}
}

/* The three pixel loops of the Rasterizer: S rays per listed pixel in the reference's loop order.
 * mode 0: Serial_Normal_Raster rasterizer.cpp:230 (S = 1); 1: Anti_Aliasing :327-343 (S = 4); 2: Depth_Of_Field_Effect
 * :378-395 (S = 49: DistLensDim 7, Lwidth 0.375, Lheight 0.3, :61-64, :71-72).  o, d: n·S × 3. */
void rto_raster_rays(int mode, const double *c, uint32_t x_res, uint32_t y_res, long n, const uint32_t *pix, double *o,
                     double *d) {
	double rd[18];
	rto_raster_details(c, x_res, y_res, rd);
	const v3 eye = ld3(c), lookat = ld3(c + 3), U = ld3(rd + 3), R = ld3(rd + 6), O = ld3(rd + 9), dR = ld3(rd + 12), dU = ld3(rd + 15);
	const double vpdist = c[9], xmin = c[10], ymax = c[13];
	const int dim = 7, grad = dim / 2;
	const v3 XG = divs(scl(0.375, R), (double)dim), YG = divs(scl(0.3, U), (double)dim);
	for (long k = 0; k < n; ++k) {
		const uint32_t i = pix[2 * k], j = pix[2 * k + 1];
		if (mode == 1) {
			long s = 0;
			for (float fy = 1; fy < 4; fy += 2)
				for (float fx = 1; fx < 4; fx += 2, ++s) {
					st3(o + 3 * (4 * k + s), eye);
					st3(d + 3 * (4 * k + s), unit(sub(add(O, scl(j + (fx * 0.25), dR)), scl(i + (fy * 0.25), dU))));
				}
		} else if (mode == 2) {
			long s = 0;
			for (int yL = grad; yL >= -grad; --yL)
				for (int xL = -grad; xL <= grad; ++xL, ++s) {
					const v3 lens = add(add(eye, scl((double)xL, XG)), scl((double)yL, YG));
					const v3 GL = unit(sub(lookat, lens));
					const v3 OL = add(add(scl(vpdist, GL), scl(xmin, R)), scl(ymax, U));
					st3(o + 3 * (49 * k + s), lens);
					st3(d + 3 * (49 * k + s), unit(sub(add(OL, scl(j + 0.5, dR)), scl(i + 0.5, dU))));
				}
		} else {
			st3(o + 3 * k, eye);
			st3(d + 3 * k, unit(sub(add(O, scl(j + 0.5, dR)), scl(i + 0.5, dU))));
		}
	}
}

/* object.cpp:23 */
static inline v3 hit_point(v3 o, v3 d, v3 v1, v3 n) {
	return add(o, divs(scl(dot(sub(v1, o), n), d), dot(n, d)));
}
void rto_hit_point(const double o[3], const double d[3], const double v1[3], const double nrm[3], double p[3]) {
	st3(p, hit_point(ld3(o), ld3(d), ld3(v1), ld3(nrm)));
}

void rto_shadow_rays(long n, const double *P, const double *N, const double C[3], double *o, double *d) {
	v3 c = ld3(C);
	for (long k = 0; k < n; ++k) {
		v3 p = ld3(P + 3 * k), nn = ld3(N + 3 * k);
		st3(d + 3 * k, unit(sub(c, p)));
		st3(o + 3 * k, add(p, scl(0.01, nn)));
	}
}

/* shader.cpp:41-103 generateSoftShadows: n_points rays per point toward the grid through the light centre (:60-71),
 * origin P + 0.05·dir (:90-91); ray k·n_points + i·divY + j */
void rto_soft_shadow_rays(long n, const double *P, const double C[3], uint32_t n_points, double *o, double *d) {
	uint32_t div_x = n_points, div_y = 1;
	while (div_x % 2 == 0) { div_x /= 2; div_y *= 2; }
	for (long k = 0; k < n; ++k) {
		v3 p = ld3(P + 3 * k);
		for (uint32_t i = 0; i < div_x; ++i)
			for (uint32_t j = 0; j < div_y; ++j) {
				v3 target = {C[0] - 1 + ((j * 2 + 1) / (double)div_y), C[1] - 1 + ((i * 2 + 1) / (double)div_x), C[2]};
				v3 dir = unit(sub(target, p));
				long at = k * (long)n_points + (long)i * div_y + j;
				st3(d + 3 * at, dir);
				st3(o + 3 * at, add(p, scl(0.05, dir)));
			}
	}
}

static inline uint64_t mix64(uint64_t z);
/* shader.cpp:140-232 findOcclusionNew: the hemisphere frame and the AO ray of one square point (a, b) in
 * [-1,1)^2 (the reference draws a, b from an unseeded mt19937, :148-150,177; here they are injected).
 * Returns direction (unit) and origin P + dir·1e-4. */
static void ao_frame(v3 normal, v3 *X, v3 *Y, v3 *Z) {
	*Z = unit(normal);                                   /* :153 */
	v3 t = *Z;
	double cz = fabs(Z->z), cx = fabs(Z->x), cy = fabs(Z->y);
	if (cz <= cx) {                                      /* :160-168, the reference's own branch structure */
		if (cz <= cy) t.z = 1;
	} else if (cx <= cz) {
		if (cx <= cy) t.x = 1;
	} else t.y = 1;
	*X = unit(cross(t, *Z));                             /* :170 */
	*Y = cross(*Z, *X);                                  /* :171 */
}
static void ao_ray(v3 P, v3 X, v3 Y, v3 Z, double R, double a, double b, v3 *o, v3 *d) {
	double phi, r;
	if (a > -b) {                                        /* :182-204 square -> disc */
		if (a > b) { r = a; phi = (RTO_PI / 4.0) * (b / a); }
		else { r = b; phi = (RTO_PI / 4.0) * (2.0 - (a / b)); }
	} else {
		if (a < b) { r = -a; phi = (RTO_PI / 4.0) * (4.0 + (b / a)); }
		else { r = -b; phi = (b != 0) ? (RTO_PI / 4.0) * (6.0 - (a / b)) : 0; }
	}
	double u = r * cos(phi), v = r * sin(phi);           /* :206-207 */
	v3 base = add(scl(u, X), scl(v, Y));                 /* :210-211 */
	double leg = (R * R) - (len(base) * len(base));      /* :212 */
	v3 dir = unit(add(scl(leg / R, Z), base));           /* :214,218 */
	*d = dir;
	*o = add(P, scl(.0001, dir));                        /* :220 */
}
void rto_ao_square_point(uint64_t seed, uint64_t ray, uint32_t sample, double ab[2]) {
	uint64_t h = mix64(mix64(seed ^ ray) ^ (uint64_t)sample);
	ab[0] = (2.0 * ((double)(h >> 11) * 0x1.0p-53)) - 1.0;          /* dis(gen) - 1, dis = uniform(0,2): :150,177 */
	ab[1] = (2.0 * ((double)(mix64(h) >> 11) * 0x1.0p-53)) - 1.0;
}
void rto_ao_rays(long n, const double *P, const double *N, double R, uint32_t spp, uint64_t seed, double *o, double *d) {
	for (long k = 0; k < n; ++k) {
		v3 X, Y, Z;
		ao_frame(ld3(N + 3 * k), &X, &Y, &Z);
		for (uint32_t s = 0; s < spp; ++s) {
			double ab[2];
			rto_ao_square_point(seed, (uint64_t)k, s, ab);
			v3 oo, dd;
			ao_ray(ld3(P + 3 * k), X, Y, Z, R, ab[0], ab[1], &oo, &dd);
			st3(o + 3 * ((long)k * spp + s), oo);
			st3(d + 3 * ((long)k * spp + s), dd);
		}
	}
}
/* same construction on injected square points: m rays, frame from N[k], point (a,b) = ab[2k..] */
void rto_ao_rays_points(long m, const double *P, const double *N, double R, const double *ab, double *o, double *d) {
	for (long k = 0; k < m; ++k) {
		v3 X, Y, Z, oo, dd;
		ao_frame(ld3(N + 3 * k), &X, &Y, &Z);
		ao_ray(ld3(P + 3 * k), X, Y, Z, R, ab[2 * k], ab[2 * k + 1], &oo, &dd);
		st3(o + 3 * k, oo);
		st3(d + 3 * k, dd);
	}
}

/* ---- radiosity -------------------------------------------------------------- */
double rto_area(const double a[3], const double b[3], const double c[3]) {
	return len(cross(sub(ld3(b), ld3(a)), sub(ld3(c), ld3(a)))) / 2.0;
}

/* The closest hit through the oracle's own hierarchy (defined below): the same record as the scan over all triangles —
 * the boxes only prune, the strict minimum and the smaller-index tie rule are kept — for inputs where the scan would
 * take hours (one row of a 64 K-patch matrix is 4·10^10 triangle tests).  NULL = the scan. */
static int32_t bvh_closest_one(const rto_bvh *b, v3 O, v3 D, double tmax, double *bt, double *pu, double *pv);
static int32_t closest_via(const rto_bvh *b, long n, const double *verts, v3 org, v3 dir, double *t, double *u, double *v) {
	return b ? bvh_closest_one(b, org, dir, RTO_INFINITY, t, u, v) : closest(n, verts, org, dir, RTO_INFINITY, t, u, v);
}

/* radiosity.cpp:231-262 */
static int visible_via(const rto_bvh *b, long n, const double *verts, const double *normals, int32_t j, v3 xi, v3 yi) {
	v3 dir = unit(sub(yi, xi));
	v3 org = add(xi, scl(RTO_EPSILON, dir));
	double t, u, v;
	int32_t h = closest_via(b, n, verts, org, dir, &t, &u, &v);
	if (h < 0) return 0;
	v3 p = hit_point(org, dir, ld3(verts + 9 * h), ld3(normals + 3 * h));
	return (len(sub(p, yi)) < RTO_EPSILON && dot(dir, ld3(normals + 3 * j)) < 0) ? 1 : 0;
}
static int visible(long n, const double *verts, const double *normals, int32_t j, v3 xi, v3 yi) {
	return visible_via(NULL, n, verts, normals, j, xi, yi);
}

void rto_visible(long n, const double *verts, const double *normals, long m, const int32_t *ei, const int32_t *ej,
                 const double *xi, const double *xj, uint8_t *vis) {
	(void)ei;
#pragma omp parallel for schedule(dynamic, 16)
	for (long k = 0; k < m; ++k) vis[k] = (uint8_t)visible(n, verts, normals, ej[k], ld3(xi + 3 * k), ld3(xj + 3 * k));
}

static int light_blocked(long n, const double *verts, v3 ni, v3 x, v3 C) {
	v3 dir = unit(sub(C, x));
	v3 org = add(x, scl(RTO_EPSILON, ni));
	double t, u, v;
	return closest(n, verts, org, dir, RTO_INFINITY, &t, &u, &v) >= 0;
}

void rto_light_blocked(long n, const double *verts, const double *normals, long m, const int32_t *ei,
                       const double *xi, const double C[3], uint8_t *blocked) {
#pragma omp parallel for schedule(dynamic, 16)
	for (long k = 0; k < m; ++k)
		blocked[k] = (uint8_t)light_blocked(n, verts, ld3(normals + 3 * ei[k]), ld3(xi + 3 * k), ld3(C));
}

/* splitmix64 finaliser chained over the stream coordinates */
static inline uint64_t mix64(uint64_t z) {
	z += 0x9E3779B97F4A7C15ull;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}
uint64_t rto_hash(uint64_t seed, uint32_t a, uint32_t b, uint32_t k, uint32_t which) {
	uint64_t h = mix64(seed ^ (((uint64_t)a << 32) | b));
	return mix64(h ^ (((uint64_t)k << 32) | which));
}

void rto_sample_point(const double *e, uint64_t seed, uint32_t a, uint32_t b, uint32_t k, uint32_t which,
                      double p[3]) {
	/* one hash per point: upper 32 bits -> u, lower 32 bits -> v, each k·2^-32 (exact in double) */
	uint64_t h = rto_hash(seed, a, b, k, which);
	double u = (double)(uint32_t)(h >> 32) * 0x1.0p-32;
	double v = (double)(uint32_t)h * 0x1.0p-32;
	if (u + v > 1.0) { u = 1.0 - u; v = 1.0 - v; }
	double w = 1.0 - u - v;
	for (int c = 0; c < 3; ++c) p[c] = (w * e[c]) + (u * e[3 + c]) + (v * e[6 + c]); /* util.cpp:113-115 */
}

/* radiosity.cpp:305-333 */
static double ff_term_via(const rto_bvh *bvh, long n, const double *verts, const double *normals, int32_t i, int32_t j, uint32_t S,
                          const double *xi, const double *xj) {
	double ff = 0;
	v3 ni = ld3(normals + 3 * i), nj = ld3(normals + 3 * j);
	double area_j = rto_area(verts + 9 * j, verts + 9 * j + 3, verts + 9 * j + 6);
	for (uint32_t k = 0; k < S; ++k) {
		v3 a = ld3(xi + 3 * k), b = ld3(xj + 3 * k);
		if (visible_via(bvh, n, verts, normals, j, a, b)) {
			v3 ab = sub(a, b);
			double dist2 = dot(ab, ab);
			double c1 = dot(unit(sub(b, a)), ni);
			double c2 = dot(unit(sub(a, b)), nj);
			double cosFactor = c1 * c2;
			double delF = (c1 < 0.0 || c2 < 0.0) ? 0.0 : cosFactor / ((RTO_PI * dist2) + (area_j / (int)S));
			if (delF > 0.0) ff += delF;
		}
	}
	ff = ff / ((double)S);
	return ff * area_j;
}

static double ff_term(long n, const double *verts, const double *normals, int32_t i, int32_t j, uint32_t S,
                      const double *xi, const double *xj) {
	return ff_term_via(NULL, n, verts, normals, i, j, S, xi, xj);
}

double rto_ff_term_points(long n, const double *verts, const double *normals, int32_t i, int32_t j, uint32_t S,
                          const double *xi, const double *xj) {
	return ff_term(n, verts, normals, i, j, S, xi, xj);
}

void rto_form_factors(long n, const double *verts, const double *normals, uint32_t S, uint64_t seed, long row0,
                      long n_rows, long ld, double *F) {
#pragma omp parallel for schedule(dynamic, 1)
	for (long r = 0; r < n_rows; ++r) {
		long i = row0 + r;
		double *xi = (double *)malloc(sizeof(double) * 6 * S), *xj = xi + 3 * S;
		for (long j = 0; j < n; ++j) {
			if (i == j) { F[r * ld + j] = 0.0; continue; }
			for (uint32_t k = 0; k < S; ++k) {
				rto_sample_point(verts + 9 * i, seed, (uint32_t)i, (uint32_t)j, k, 0, xi + 3 * k);
				rto_sample_point(verts + 9 * j, seed, (uint32_t)i, (uint32_t)j, k, 1, xj + 3 * k);
			}
			F[r * ld + j] = ff_term(n, verts, normals, (int32_t)i, (int32_t)j, S, xi, xj);
		}
		free(xi);
	}
}

/* The same rows with every closest hit taken through the oracle's hierarchy, columns spread over the threads
 * (tests/test_oracle_golden.py pins it to rto_form_factors on small scenes). */
void rto_form_factors_bvh(const rto_bvh *bvh, long n, const double *verts, const double *normals, uint32_t S, uint64_t seed,
                          long row0, long n_rows, long ld, double *F) {
	for (long r = 0; r < n_rows; ++r) {
		long i = row0 + r;
#pragma omp parallel
		{
			double *xi = (double *)malloc(sizeof(double) * 6 * S), *xj = xi + 3 * S;
#pragma omp for schedule(dynamic, 64)
			for (long j = 0; j < n; ++j) {
				if (i == j) { F[r * ld + j] = 0.0; continue; }
				for (uint32_t k = 0; k < S; ++k) {
					rto_sample_point(verts + 9 * i, seed, (uint32_t)i, (uint32_t)j, k, 0, xi + 3 * k);
					rto_sample_point(verts + 9 * j, seed, (uint32_t)i, (uint32_t)j, k, 1, xj + 3 * k);
				}
				F[r * ld + j] = ff_term_via(bvh, n, verts, normals, (int32_t)i, (int32_t)j, S, xi, xj);
			}
			free(xi);
		}
	}
}

/* radiosity.cpp:264-303 */
static double light_term(long n, const double *verts, const double *normals, v3 C, int32_t i, uint32_t S,
                         const double *xi, int mirror_bug) {
	double ff = 0.0;
	v3 ni = ld3(normals + 3 * i);
	for (uint32_t k = 0; k < S; ++k) {
		if (mirror_bug) ff = 0.0; /* :276 */
		v3 x = ld3(xi + 3 * k);
		if (!light_blocked(n, verts, ni, x, C)) {
			v3 dir = unit(sub(C, x));
			double L = len(sub(C, x));
			double tmp = ((1.0 / (4.0 * RTO_PI)) * (dot(dir, ni) / (L * L)));
			if (tmp > 0.0) ff += tmp;
		}
	}
	return ff / ((double)S);
}

double rto_light_term_points(long n, const double *verts, const double *normals, const double C[3], int32_t i,
                             uint32_t S, const double *xi, int mirror_bug) {
	return light_term(n, verts, normals, ld3(C), i, S, xi, mirror_bug);
}

void rto_light_terms(long n, const double *verts, const double *normals, const double C[3], uint32_t S,
                     uint64_t seed, int mirror_bug, long row0, long n_rows, double *f) {
#pragma omp parallel for schedule(dynamic, 4)
	for (long r = 0; r < n_rows; ++r) {
		long i = row0 + r;
		double *xi = (double *)malloc(sizeof(double) * 3 * S);
		for (uint32_t k = 0; k < S; ++k)
			rto_sample_point(verts + 9 * i, seed, (uint32_t)i, 0xFFFFFFFFu, k, 0, xi + 3 * k);
		f[r] = light_term(n, verts, normals, ld3(C), (int32_t)i, S, xi, mirror_bug);
		free(xi);
	}
}

/* radiosity.cpp:469-481 on K = I − ρF (element rows), light row = identity (:413-428) */
static int converged(long n, const double *F, double rho, const double *B, const double *E) {
	for (long i = 0; i < n; ++i) {
		double sum = 0.0;
		for (long j = 0; j < n; ++j) {
			double I = (i == j) ? 1.0 : 0.0;
			double Kij = (i < n - 1) ? I - (rho * F[i * n + j]) : I - F[i * n + j];
			sum += Kij * B[j];
		}
		if (fabs(sum - E[i]) > RTO_EPSILON) return 0;
	}
	return 1;
}

long rto_progressive_refinement(long n, const double *F, const double *area, double rho, double power,
                                long max_shots, double *B) {
	double *E = (double *)calloc((size_t)n, sizeof(double));
	double *dB = (double *)malloc(sizeof(double) * (size_t)n);
	double *tmp = (double *)calloc((size_t)n, sizeof(double)); /* [n-1] stays 0: see header */
	E[n - 1] = power;
	for (long i = 0; i < n; ++i) { dB[i] = E[i]; B[i] = E[i]; }
	int first = 1;
	long shots = 0;
	while (!converged(n, F, rho, B, E)) {
		if (shots >= max_shots) { shots = -1; break; }
		long indx;
		if (first) {
			first = 0;
			for (long i = 0; i < n - 1; ++i) {
				double dRad = dB[n - 1] * F[i * n + (n - 1)] * rho;
				dB[i] += dRad;
				B[i] += dRad;
			}
			indx = n - 1;
		} else {
			for (long i = 0; i < n - 1; ++i) tmp[i] = dB[i] * area[i];
			double val = 0.0; /* util.cpp:84-94 */
			indx = 0;
			for (long i = 0; i < n; ++i)
				if (val < tmp[i]) { val = tmp[i]; indx = i; }
			for (long i = 0; i < n; ++i) {
				double dRad = dB[indx] * F[i * n + indx] * rho;
				dB[i] += dRad;
				B[i] += dRad;
			}
		}
		dB[indx] = 0.0;
		++shots;
	}
	free(E); free(dB); free(tmp);
	return shots;
}

int rto_direct_solve(long n, const double *F, double rho, double power, double *B) {
	double *A = (double *)malloc(sizeof(double) * (size_t)n * (size_t)(n + 1));
	long w = n + 1;
	for (long i = 0; i < n; ++i) {
		for (long j = 0; j < n; ++j) {
			double I = (i == j) ? 1.0 : 0.0;
			A[i * w + j] = (i < n - 1) ? I - rho * F[i * n + j] : I - F[i * n + j];
		}
		A[i * w + n] = (i == n - 1) ? power : 0.0;
	}
	for (long c = 0; c < n; ++c) {
		long piv = c;
		for (long r = c + 1; r < n; ++r)
			if (fabs(A[r * w + c]) > fabs(A[piv * w + c])) piv = r;
		if (A[piv * w + c] == 0.0) { free(A); return 1; }
		if (piv != c)
			for (long j = 0; j <= n; ++j) { double t = A[c * w + j]; A[c * w + j] = A[piv * w + j]; A[piv * w + j] = t; }
		for (long r = c + 1; r < n; ++r) {
			double f = A[r * w + c] / A[c * w + c];
			if (f == 0.0) continue;
			for (long j = c; j <= n; ++j) A[r * w + j] -= f * A[c * w + j];
		}
	}
	for (long i = n - 1; i >= 0; --i) {
		double s = A[i * w + n];
		for (long j = i + 1; j < n; ++j) s -= A[i * w + j] * B[j];
		B[i] = s / A[i * w + i];
	}
	free(A);
	return 0;
}

long rto_jacobi_rgb(long n, const float *F, const double rho[3], const double *E, double tol_rel, long max_sweeps,
                    double *B) {
	double *Bn = (double *)malloc(sizeof(double) * 3 * (size_t)n);
	memcpy(B, E, sizeof(double) * 3 * (size_t)n);
	long s = 0;
	for (; s < max_sweeps; ++s) {
		double d2 = 0, b2 = 0;
#pragma omp parallel for reduction(+ : d2, b2) schedule(static)
		for (long i = 0; i < n; ++i) {
			double a0 = 0, a1 = 0, a2 = 0;
			const float *row = F + (size_t)i * (size_t)n;
			for (long j = 0; j < n; ++j) {
				double f = (double)row[j];
				a0 += f * B[3 * j]; a1 += f * B[3 * j + 1]; a2 += f * B[3 * j + 2];
			}
			double r[3] = {E[3 * i] + rho[0] * a0, E[3 * i + 1] + rho[1] * a1, E[3 * i + 2] + rho[2] * a2};
			for (int c = 0; c < 3; ++c) {
				double dd = r[c] - B[3 * i + c];
				d2 += dd * dd;
				b2 += r[c] * r[c];
				Bn[3 * i + c] = r[c];
			}
		}
		memcpy(B, Bn, sizeof(double) * 3 * (size_t)n);
		if (sqrt(d2) <= tol_rel * sqrt(b2)) { ++s; break; }
	}
	free(Bn);
	return s;
}

/* ---- CPU baseline BVH (ours; the reference's own accelerator is a BSP that cannot build
 * the large synthetic scenes) ------------------------------------------ */
typedef struct { double lo[3], hi[3]; int32_t left, right, first, count; } bnode;
struct rto_bvh { long n_tris, n_nodes; bnode *nodes; int32_t *order; const double *tris; double *tcopy; };

typedef struct { double lo[3], hi[3], c[3]; } tbox;

static void box_of(const double *p, tbox *b) {
	for (int a = 0; a < 3; ++a) {
		double x = p[a], y = p[3 + a], z = p[6 + a];
		double lo = x < y ? x : y; lo = lo < z ? lo : z;
		double hi = x > y ? x : y; hi = hi > z ? hi : z;
		b->lo[a] = lo; b->hi[a] = hi; b->c[a] = 0.5 * (lo + hi);
	}
}
static double half_area(const double lo[3], const double hi[3]) {
	double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
	return dx * dy + dy * dz + dz * dx;
}

static int32_t build_rec(rto_bvh *b, const tbox *tb, int32_t first, int32_t count, int depth) {
	int32_t id = (int32_t)b->n_nodes++;
	bnode *nd = &b->nodes[id];
	double clo[3] = {1e300, 1e300, 1e300}, chi[3] = {-1e300, -1e300, -1e300};
	for (int a = 0; a < 3; ++a) { nd->lo[a] = 1e300; nd->hi[a] = -1e300; }
	for (int32_t i = first; i < first + count; ++i) {
		const tbox *t = &tb[b->order[i]];
		for (int a = 0; a < 3; ++a) {
			if (t->lo[a] < nd->lo[a]) nd->lo[a] = t->lo[a];
			if (t->hi[a] > nd->hi[a]) nd->hi[a] = t->hi[a];
			if (t->c[a] < clo[a]) clo[a] = t->c[a];
			if (t->c[a] > chi[a]) chi[a] = t->c[a];
		}
	}
	nd->left = nd->right = -1; nd->first = first; nd->count = count;
	if (count <= 4 || depth > 60) return id;
	enum { NB = 16 };
	int best_axis = -1, best_split = 0;
	double best_cost = 1e300;
	for (int a = 0; a < 3; ++a) {
		double ext = chi[a] - clo[a];
		if (!(ext > 0)) continue;
		int cnt[NB] = {0};
		double blo[NB][3], bhi[NB][3];
		for (int k = 0; k < NB; ++k)
			for (int c = 0; c < 3; ++c) { blo[k][c] = 1e300; bhi[k][c] = -1e300; }
		for (int32_t i = first; i < first + count; ++i) {
			const tbox *t = &tb[b->order[i]];
			int k = (int)(NB * ((t->c[a] - clo[a]) / ext));
			if (k >= NB) k = NB - 1;
			cnt[k]++;
			for (int c = 0; c < 3; ++c) {
				if (t->lo[c] < blo[k][c]) blo[k][c] = t->lo[c];
				if (t->hi[c] > bhi[k][c]) bhi[k][c] = t->hi[c];
			}
		}
		double ra[NB]; int rc[NB];
		double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
		int c2 = 0;
		for (int k = NB - 1; k > 0; --k) {
			for (int c = 0; c < 3; ++c) { if (blo[k][c] < lo[c]) lo[c] = blo[k][c]; if (bhi[k][c] > hi[c]) hi[c] = bhi[k][c]; }
			c2 += cnt[k];
			ra[k] = c2 ? half_area(lo, hi) : 0; rc[k] = c2;
		}
		for (int c = 0; c < 3; ++c) { lo[c] = 1e300; hi[c] = -1e300; }
		int c1 = 0;
		for (int k = 0; k < NB - 1; ++k) {
			for (int c = 0; c < 3; ++c) { if (blo[k][c] < lo[c]) lo[c] = blo[k][c]; if (bhi[k][c] > hi[c]) hi[c] = bhi[k][c]; }
			c1 += cnt[k];
			if (c1 == 0 || rc[k + 1] == 0) continue;
			double cost = half_area(lo, hi) * c1 + ra[k + 1] * rc[k + 1];
			if (cost < best_cost) { best_cost = cost; best_axis = a; best_split = k; }
		}
	}
	int32_t mid;
	if (best_axis < 0) {
		mid = first + count / 2;
	} else {
		double ext = chi[best_axis] - clo[best_axis];
		int32_t i = first, j = first + count - 1;
		while (i <= j) {
			const tbox *t = &tb[b->order[i]];
			int k = (int)(NB * ((t->c[best_axis] - clo[best_axis]) / ext));
			if (k >= NB) k = NB - 1;
			if (k <= best_split) ++i;
			else { int32_t s = b->order[i]; b->order[i] = b->order[j]; b->order[j] = s; --j; }
		}
		mid = i;
		if (mid == first || mid == first + count) mid = first + count / 2;
	}
	int32_t l = build_rec(b, tb, first, mid - first, depth + 1);
	int32_t r = build_rec(b, tb, mid, first + count - mid, depth + 1);
	b->nodes[id].left = l; b->nodes[id].right = r;
	return id;
}

rto_bvh *rto_bvh_build(long n, const double *tris) {
	rto_bvh *b = (rto_bvh *)calloc(1, sizeof(rto_bvh));
	b->n_tris = n;
	b->tcopy = (double *)malloc(sizeof(double) * 9 * (size_t)n);
	memcpy(b->tcopy, tris, sizeof(double) * 9 * (size_t)n);
	b->tris = b->tcopy;
	b->nodes = (bnode *)malloc(sizeof(bnode) * (size_t)(2 * n + 1));
	b->order = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
	tbox *tb = (tbox *)malloc(sizeof(tbox) * (size_t)n);
	for (long i = 0; i < n; ++i) { b->order[i] = (int32_t)i; box_of(tris + 9 * i, &tb[i]); }
	if (n > 0) build_rec(b, tb, 0, (int32_t)n, 0);
	free(tb);
	return b;
}

void rto_bvh_free(rto_bvh *b) {
	if (!b) return;
	free(b->nodes); free(b->order); free(b->tcopy); free(b);
}

/* slab test in double, widened by a relative 1e-9 so the exact test decides */
static inline int box_hit(const bnode *n, v3 o, v3 inv, double tbest, double *tn) {
	double pad = 1e-9 * (fabs(o.x) + fabs(o.y) + fabs(o.z) + fabs(n->hi[0]) + fabs(n->lo[0]) + fabs(n->hi[1]) +
	                     fabs(n->lo[1]) + fabs(n->hi[2]) + fabs(n->lo[2]) + 1.0);
	double t0 = 0, t1 = tbest * (1 + 1e-9) + 1e-12;
	const double oo[3] = {o.x, o.y, o.z}, ii[3] = {inv.x, inv.y, inv.z};
	for (int a = 0; a < 3; ++a) {
		double ta = (n->lo[a] - pad - oo[a]) * ii[a], tb = (n->hi[a] + pad - oo[a]) * ii[a];
		if (isnan(ta) || isnan(tb)) continue; /* 0 * inf: origin on a padded plane of a parallel slab */
		if (ta > tb) { double s = ta; ta = tb; tb = s; }
		if (ta > t0) t0 = ta;
		if (tb < t1) t1 = tb;
	}
	*tn = t0;
	return t0 <= t1;
}

/* one ray through the BVH: same winner as the in-order scan (smaller t, or equal t and smaller index) */
static int32_t bvh_closest_one(const rto_bvh *b, v3 O, v3 D, double tmax, double *bt, double *pu, double *pv) {
	v3 inv = {1.0 / D.x, 1.0 / D.y, 1.0 / D.z};
	double best = tmax, bu = 0, bv = 0;
	int32_t bi = -1;
	int32_t stack[128];
	int sp = 0;
	if (b->n_tris > 0) stack[sp++] = 0;
	while (sp) {
		const bnode *n = &b->nodes[stack[--sp]];
		double tn;
		if (!box_hit(n, O, inv, best, &tn)) continue;
		if (n->left < 0) {
			for (int32_t i = n->first; i < n->first + n->count; ++i) {
				int32_t id = b->order[i];
				const double *p = b->tris + 9 * (long)id;
				double uu = 0, vv = 0;
				double tt = mt(O, D, ld3(p), ld3(p + 3), ld3(p + 6), &uu, &vv);
				if (tt > 0.0 && (best > tt || (bi >= 0 && best == tt && id < bi))) { best = tt; bi = id; bu = uu; bv = vv; }
			}
		} else {
			double t0, t1;
			int h0 = box_hit(&b->nodes[n->left], O, inv, best, &t0);
			int h1 = box_hit(&b->nodes[n->right], O, inv, best, &t1);
			if (h0 && h1) {
				if (t0 <= t1) { stack[sp++] = n->right; stack[sp++] = n->left; }
				else { stack[sp++] = n->left; stack[sp++] = n->right; }
			} else if (h0) stack[sp++] = n->left;
			else if (h1) stack[sp++] = n->right;
		}
	}
	*bt = bi >= 0 ? best : 0.0; *pu = bu; *pv = bv;
	return bi;
}

double rto_bvh_closest_hit(const rto_bvh *b, long n_rays, const double *o, const double *d, const double *tmax,
                           int32_t *tri, double *t, double *u, double *v) {
	struct timespec a, z;
	clock_gettime(CLOCK_MONOTONIC, &a);
#pragma omp parallel for schedule(dynamic, 256)
	for (long k = 0; k < n_rays; ++k)
		tri[k] = bvh_closest_one(b, ld3(o + 3 * k), ld3(d + 3 * k), tmax ? tmax[k] : RTO_INFINITY, &t[k], &u[k], &v[k]);
	clock_gettime(CLOCK_MONOTONIC, &z);
	return (double)(z.tv_sec - a.tv_sec) + 1e-9 * (double)(z.tv_nsec - a.tv_nsec);
}

int rto_num_threads(void) {
#ifdef _OPENMP
	return omp_get_max_threads();
#else
	return 1;
#endif
}

/* ---- Scene::Trace + Shader::Shade (scene.cpp:127-155, shader.cpp:395-645), the branch the shipped
 * main.cpp runs (scene.radiosity == NULL) ------------------------------------------------------- */
typedef struct {
	long n_tris;
	const double *tris, *normals, *mats, *lights;
	const uint32_t *tri_object;
	int n_lights;
	const rto_bvh *bvh;
	uint32_t max_depth, ao_rays;
	double ao_radius;
	uint64_t seed;
	uint32_t flags; /* RTO_SHADE_RADIOSITY: Shade's radiosity branch (scene.radiosity != NULL) */
} shade_ctx;

static const v3 kBackground = {0.15, 0.25, 0.35}; /* scene.cpp:21 */
static const v3 kTermination = {0.0, 0.0, 0.0};   /* scene.cpp:22 */

static inline v3 cmul(v3 a, v3 b) { v3 r = {a.x * b.x, a.y * b.y, a.z * b.z}; return r; } /* color.h:43 */
static inline double max2(double x, double y) { return x >= y ? x : y; }                /* util.h:26 */
static inline double min3(double x, double y, double z) {                              /* util.cpp:7 */
	if (x <= y && x <= z) return x;
	if (y <= z) return y;
	return z;
}

static int32_t cast(const shade_ctx *c, v3 O, v3 D, double tmax, double *t) {
	double u, v;
	if (c->bvh) return bvh_closest_one(c->bvh, O, D, tmax, t, &u, &v);
	return closest(c->n_tris, c->tris, O, D, tmax, t, &u, &v);
}

/* findOcclusionNew (shader.cpp:140-232); square points of stream (mix64(seed ^ path), k0, sample) */
static double occlusion(const shade_ctx *c, v3 P, v3 normal, uint64_t k0, uint64_t path) {
	if (c->ao_rays == 0) return 1;                       /* :142 */
	int occ = 0;
	v3 X, Y, Z;
	ao_frame(normal, &X, &Y, &Z);
	const uint64_t node_seed = mix64(c->seed ^ path);
	for (uint32_t s = 0; s < c->ao_rays; ++s) {
		double ab[2], t;
		v3 o, d;
		rto_ao_square_point(node_seed, k0, s, ab);
		ao_ray(P, X, Y, Z, c->ao_radius, ab[0], ab[1], &o, &d);
		if (cast(c, o, d, c->ao_radius, &t) >= 0) occ++;  /* :226  Cast ∧ distance < R */
	}
	return (1.0 - (occ / (double)c->ao_rays));           /* :231 */
}

static v3 trace_rec(const shade_ctx *c, v3 O, v3 D, uint32_t generation, uint64_t k0, uint64_t path);

static v3 shade(const shade_ctx *c, v3 O, v3 D, int32_t tri, double distance, uint32_t generation, uint64_t k0,
                uint64_t path) {
	const double *m = c->mats + 18 * (size_t)(c->tri_object ? c->tri_object[tri] : 0);
	const v3 diffuse = ld3(m), specular = ld3(m + 3), ambient = ld3(m + 6), reflective = ld3(m + 9);
	v3 transl = ld3(m + 12);
	const double e = m[15], k = m[16], rough = m[17];
	const v3 Ns = ld3(c->normals + 3 * (size_t)tri);     /* hit.normal = triNormal, object.cpp:17 */
	const v3 P = hit_point(O, D, ld3(c->tris + 9 * (size_t)tri), Ns);
	const v3 N = unit(Ns);                               /* :416 */
	const v3 E = unit(sub(O, P));                        /* :417 */
	v3 R;
	v3 color = {0, 0, 0};                                /* :427-429 */

	if (c->flags & RTO_SHADE_RADIOSITY) {
		/* shader.cpp:431-478 as shipped: radiosityAmbient (trace_ray, :433) and occlusion (findOcclusionNew with 100 rays, :434)
		 * are computed and never used — the lines that would use them are commented out (:457-465) — so the colour is the
		 * specular sum over the lights (:467-472), returned before shadows, reflection and refraction (:477). */
		for (int i = 0; i < c->n_lights; ++i) {
			const v3 LightPos = ld3(c->lights + 6 * i), emission = ld3(c->lights + 6 * i + 3);
			const v3 L = unit(sub(LightPos, P));             /* :453 */
			const v3 RR = add(scl(-1.0, L), scl(2.0 * dot(L, N), N)); /* :467 */
			color = add(color, cmul(emission, scl(pow(max2(0.0, dot(E, RR)), e), specular))); /* :472 */
		}
		return color;
	}
	const double occ = occlusion(c, P, Ns, k0, path);    /* :481 */
	for (int i = 0; i < c->n_lights; ++i) {              /* :485-519 */
		const v3 LightPos = ld3(c->lights + 6 * i), emission = ld3(c->lights + 6 * i + 3);
		const v3 L = unit(sub(LightPos, P));
		color = add(color, cmul(emission, add(scl(.85 * occ, ambient), scl(max2(0, dot(N, L)), diffuse)))); /* :494 */
		if (rough != 0.0) {                              /* :496-508 */
			const v3 H = unit(add(L, E));
			const double LH = dot(L, H), NH = dot(N, H);
			const double g = sqrt((30.0 * 3016.0) - 1.0 + (LH * LH));
			double Fresnel = 0.5 * pow(g - LH, 2) / pow(g + LH, 2);
			Fresnel *= pow(LH * (g + LH) - 1.0, 2) / pow(LH * (g - LH) + 1.0, 2) + 1.0;
			const double mm = 0.4;
			double Dm = 1.0 / (4.0 * mm * mm * pow(NH, 4));
			const double DPow = (pow(NH, 2) - 1.0) / ((mm * mm) * pow(NH, 2));
			Dm = pow(Dm, DPow);
			const double G = min3(1.0, (2.0 * NH * NH) / LH, (2.0 * NH * dot(N, L)) / LH);
			const double kk = .7;
			color = add(color, scl((Fresnel * Dm * G) / (RTO_PI * dot(N, E) * dot(N, L)), scl(kk, specular)));
		} else {                                         /* :510-518 */
			const v3 RR = add(scl(-1.0, L), scl(2.0 * dot(L, N), N));
			color = add(color, cmul(emission, scl(pow(max2(0.0, dot(E, RR)), e), specular)));
		}
	}
	{                                                    /* generateNormalShadows, :105-138 */
		const v3 dir = unit(sub(ld3(c->lights), P));
		const v3 org = add(P, scl(0.01, Ns));
		double t;
		if (cast(c, org, dir, RTO_INFINITY, &t) >= 0) color = scl(0.7, ambient);
	}

	R = add(scl(-1.0, E), scl(2.0 * dot(E, Ns), Ns));    /* :532 */
	if (transl.z > 0.0 && transl.y > 0.0 && transl.x > 0.0) { /* :542 */
		double refractionCoef = 0;
		v3 tRefr = {0, 0, 0};
		const v3 mE = scl(-1, E), mN = scl(-1, Ns);
		if (dot(E, Ns) < 0.0) R = add(scl(-1.0, E), scl(2.0 * dot(E, mN), mN)); /* :571-574 */
		else R = add(scl(-1.0, E), scl(2.0 * dot(E, Ns), Ns));
		if (dot(mE, Ns) < 0.0) {                         /* :576-588 */
			const double dn = dot(mE, Ns);
			const v3 cosRefract = divs(sub(mE, scl(dn, Ns)), k);
			const double cosSq = 1 - ((1 - (dn * dn)) / (k * k));
			tRefr = sub(cosRefract, scl(sqrt(cosSq), Ns));
			refractionCoef = dot(E, Ns);
			transl.x = transl.y = transl.z = 1;
		} else {                                         /* :589-625 */
			transl.x = exp(-1.0 * transl.x * distance);
			transl.y = exp(-1.0 * transl.y * distance);
			transl.z = exp(-1.0 * transl.z * distance);
			const double dn = dot(mE, mN);
			const v3 cosRefract = divs(sub(mE, scl(dn, mN)), k);
			const double cosSq = 1 - ((1 - (dn * dn)) / (k * k));
			if (cosSq < 0.0)                             /* :608-615 total internal reflection */
				return cmul(transl, trace_rec(c, add(P, scl(RTO_EPSILON, R)), R, generation + 1, k0, 2 * path));
			tRefr = sub(cosRefract, scl(sqrt(cosSq), mN));
			refractionCoef = dot(tRefr, Ns);
		}
		const double beerR0 = ((1 - k) * (1 - k)) / ((1 + k) * (1 + k));          /* :627 */
		const double beerR = beerR0 + (1 - beerR0) * pow(1 - refractionCoef, 5);  /* :628 */
		const v3 a = trace_rec(c, add(P, scl(RTO_EPSILON, R)), R, generation + 1, k0, 2 * path);
		const v3 b = trace_rec(c, add(P, scl(RTO_EPSILON, tRefr)), tRefr, generation + 1, k0, 2 * path + 1);
		return cmul(transl, add(scl(beerR, a), scl(1 - beerR, b)));               /* :640 */
	}
	return add(color, cmul(reflective, trace_rec(c, add(P, scl(RTO_EPSILON, R)), R, generation + 1, k0, 2 * path))); /* :643 */
}

static v3 trace_rec(const shade_ctx *c, v3 O, v3 D, uint32_t generation, uint64_t k0, uint64_t path) {
	if (generation > c->max_depth) return kTermination;  /* scene.cpp:134-137 */
	double t;
	const int32_t tri = cast(c, O, D, RTO_INFINITY, &t);
	if (tri < 0) return kBackground;                     /* scene.cpp:151 */
	return shade(c, O, D, tri, t, generation, k0, path);
}

void rto_trace(long n_tris, const double *tris, const double *normals, const uint32_t *tri_object, const double *mats18,
               const double *lights6, int n_lights, const rto_bvh *bvh, uint32_t max_tree_depth, uint32_t ao_rays,
               double ao_radius, uint64_t seed, long n_rays, const double *o, const double *d, uint32_t generation,
               uint64_t first_index, double *rgb) {
	rto_trace_flags(n_tris, tris, normals, tri_object, mats18, lights6, n_lights, bvh, max_tree_depth, ao_rays, ao_radius, seed, 0u,
	                n_rays, o, d, generation, first_index, rgb);
}

void rto_trace_flags(long n_tris, const double *tris, const double *normals, const uint32_t *tri_object, const double *mats18,
                     const double *lights6, int n_lights, const rto_bvh *bvh, uint32_t max_tree_depth, uint32_t ao_rays,
                     double ao_radius, uint64_t seed, uint32_t flags, long n_rays, const double *o, const double *d,
                     uint32_t generation, uint64_t first_index, double *rgb) {
	shade_ctx c = {n_tris, tris, normals, mats18, lights6, tri_object, n_lights, bvh, max_tree_depth, ao_rays,
	               ao_radius, seed, flags};
#pragma omp parallel for schedule(dynamic, 64)
	for (long k = 0; k < n_rays; ++k)
		st3(rgb + 3 * k, trace_rec(&c, ld3(o + 3 * k), ld3(d + 3 * k), generation, first_index + (uint64_t)k, 1));
}

/* ToneMap, rasterizer.cpp:79-88 */
void rto_tone_map(long n, const double *rgb, uint8_t *out) {
	for (long k = 0; k < 3 * n; ++k) {
		int c = (int)floor(256 * rgb[k]);
		out[k] = (uint8_t)(c >= 255 ? 255 : c);
	}
}

/* ---- radiosity display path: Rasterizer::Radiosity_Raster (rasterizer.cpp:408-470) ------------------------- */
/* calculatePseudoInverse (mat3x2.h:187-283) of the 3×2 matrix [v2−v1 | v3−v1] as returnUVofTriangle sets it up
 * (util.cpp:126-136): rows A, B with u = A·(p−v1), v = B·(p−v1).  Returns 0 when the reference reports "not invertible". */
static int rto_pinv_m(double m00, double m01, double m10, double m11, double m20, double m21, double A[3], double B[3]) {
	A[0] = A[1] = A[2] = B[0] = B[1] = B[2] = 0.0;
	const double x = m00 * m00 + m10 * m10 + m20 * m20;
	const double z = m01 * m00 + m11 * m10 + m21 * m20;
	const double y = m00 * m01 + m10 * m11 + m20 * m21;
	const double w = m01 * m01 + m11 * m11 + m21 * m21;
	double e1 = (x + w) + sqrt(pow((x + w), 2) - (4.0 * ((x * w) - (y * z))));
	e1 /= 2.0;
	double e2 = (x + w) - sqrt(pow((x + w), 2) - (4.0 * ((x * w) - (y * z))));
	e2 /= 2.0;
	if (e1 == 0.0 || e2 == 0.0) return 0;
	if (e1 == e2) {
		A[0] = m00 / e1; A[1] = m10 / e1; A[2] = m20 / e1;
		B[0] = m01 / e1; B[1] = m11 / e1; B[2] = m21 / e1;
		return 1;
	}
	if (z == 0.0 && y == 0.0) {
		A[0] = m00 / e1; A[1] = m10 / e1; A[2] = m20 / e1;
		if (fabs((A[0] * m00 + A[1] * m10 + A[2] * m20) - 1.0) > RTO_EPSILON) {
			A[0] = m00 / e2; A[1] = m10 / e2; A[2] = m20 / e2;
			B[0] = m01 / e1; B[1] = m11 / e1; B[2] = m21 / e1;
			return 1;
		}
		B[0] = m01 / e2; B[1] = m11 / e2; B[2] = m21 / e2;
		return 1;
	}
	const double l1x2 = fabs(x - e1), l1x1 = ((-1.0 * y) / (x - e1)) * l1x2;
	const double l2x2 = fabs(x - e2), l2x1 = ((-1.0 * y) / (x - e2)) * l2x2;
	double d1 = (l1x1 * l1x1) + (l1x2 * l1x2), d2 = (l2x1 * l2x1) + (l2x2 * l2x2);   /* vec2.h Unit */
	const double v1x = d1 > 0.0 ? l1x1 / sqrt(d1) : 0, v1y = d1 > 0.0 ? l1x2 / sqrt(d1) : 0;
	const double v2x = d2 > 0.0 ? l2x1 / sqrt(d2) : 0, v2y = d2 > 0.0 ? l2x2 / sqrt(d2) : 0;
	v3 U1 = {m00 * v1x + m01 * v1y, m10 * v1x + m11 * v1y, m20 * v1x + m21 * v1y};
	v3 U2 = {m00 * v2x + m01 * v2y, m10 * v2x + m11 * v2y, m20 * v2x + m21 * v2y};
	U1 = unit(divs(U1, sqrt(e1)));
	U2 = unit(divs(U2, sqrt(e2)));
	const double i1x = v1x * (1.0 / sqrt(e1)), i1y = v2x * (1.0 / sqrt(e2));
	const double i2x = v1y * (1.0 / sqrt(e1)), i2y = v2y * (1.0 / sqrt(e2));
	A[0] = i1x * U1.x + i1y * U2.x; A[1] = i1x * U1.y + i1y * U2.y; A[2] = i1x * U1.z + i1y * U2.z;
	B[0] = i2x * U1.x + i2y * U2.x; B[1] = i2x * U1.y + i2y * U2.y; B[2] = i2x * U1.z + i2y * U2.z;
	return 1;
}
int rto_pinv(const double t[9], double A[3], double B[3]) {
	return rto_pinv_m(t[3] - t[0], t[6] - t[0], t[4] - t[1], t[7] - t[1], t[5] - t[2], t[8] - t[2], A, B);
}

/* centerOfTriangle (radiosity_helper.h:115-142) */
static v3 rto_center(const double *t) {
	v3 v1 = {t[0], t[1], t[2]}, v2 = {t[3], t[4], t[5]}, v3_ = {t[6], t[7], t[8]};
	v3 l1 = sub(add(scl(0.5, sub(v3_, v2)), v2), v1);
	v3 l2 = sub(add(scl(0.5, sub(v1, v3_)), v3_), v2);
	double A[3], B[3];
	rto_pinv_m(l1.x, -l2.x, l1.y, -l2.y, l1.z, -l2.z, A, B);
	v3 d = sub(v2, v1);
	const double tt = A[0] * d.x + A[1] * d.y + A[2] * d.z;
	return add(scl(tt, l1), v1);
}
/* The vertex values of returnQuadTreeColor (radiosity_helper.h:172-213, :244-257) with nextAdj = "elements that have the same
 * three doubles as a vertex" (the reference's own adjacency build is disabled, radiosity.cpp:167-199), adjacent elements
 * taken in element order.  Plain O(n²) scan: the checker of rtmi_radiosity_vertex_values. */
void rto_vertex_values(long n, const double *verts, const double *Bv, double *out) {
	for (long e = 0; e < n; ++e) {
		const v3 ce = rto_center(verts + 9 * e);
		for (int k = 0; k < 3; ++k) {
			const double *p = verts + 9 * e + 3 * k;
			const v3 P = {p[0], p[1], p[2]};
			double acc = 0.0;
			long cnt = 0;
			for (long a = 0; a < n; ++a) {
				if (a == e) continue;
				int shares = 0;
				for (int q = 0; q < 3 && !shares; ++q) {
					const double *w = verts + 9 * a + 3 * q;
					shares = w[0] == p[0] && w[1] == p[1] && w[2] == p[2];
				}
				if (!shares) continue;
				const v3 ca = rto_center(verts + 9 * a);
				const double l1 = len(sub(ce, P)), l2 = len(sub(P, ca)), full = l1 + l2;
				acc += ((1.0 - (l1 / full)) * Bv[e]) + ((1.0 - (l2 / full)) * Bv[a]);
				++cnt;
			}
			out[3 * e + k] = cnt ? acc / cnt : Bv[e];
		}
	}
}

static double area3(v3 a, v3 b, v3 c) { return len(cross(sub(b, a), sub(c, a))) / 2.0; } /* vec3.h:116-118 */

/* Radiosity_Helper::trace_ray (radiosity_helper.cpp:318-365) + returnQuadTreeColor (radiosity_helper.h:156-268) with the
 * radiosity of each element's three vertices injected (vertex_rad n×3: what the first half of returnQuadTreeColor
 * attaches to triVert1..3; equal to the element's radiosityValue where no shared-vertex average applies).
 * Every element is tested on the whole LINE (t of either sign), inside ⇔ u ≥ 0 ∧ v ≥ 0 ∧ u+v ≤ 1 with (u,v) from the
 * pseudo-inverse, the nearest |P − o| wins (first on ties; a distance of exactly 0 counts as "none yet", :342).
 * No element → Green (0,1,0); back face (n·d > 0) → Black. */
void rto_radiosity_trace(long n_el, const double *verts, const double *normals, const double *vertex_rad, long n_rays,
                         const double *o, const double *d, double *rgb) {
	double *A = (double *)malloc(sizeof(double) * 3 * (size_t)n_el), *B = (double *)malloc(sizeof(double) * 3 * (size_t)n_el);
	int *inv = (int *)malloc(sizeof(int) * (size_t)n_el);
	for (long i = 0; i < n_el; ++i) inv[i] = rto_pinv(verts + 9 * i, A + 3 * i, B + 3 * i);
#pragma omp parallel for schedule(dynamic, 64)
	for (long k = 0; k < n_rays; ++k) {
		const v3 O = ld3(o + 3 * k), D = ld3(d + 3 * k);
		double dist = 0.0;
		long hit = -1;
		v3 fp = {0, 0, 0};
		for (long i = 0; i < n_el; ++i) {
			const v3 n = ld3(normals + 3 * i), v1 = ld3(verts + 9 * i);
			const double denom = dot(n, D);
			if (denom != 0.0) {
				const double t = -1.0 * ((dot(n, O) + dot(scl(-1.0, n), v1)) / denom);
				const v3 P = add(scl(t, D), O);
				double u = 0.0, v = 0.0;
				if (inv[i]) {
					const v3 q = sub(P, v1);
					u = A[3 * i] * q.x + A[3 * i + 1] * q.y + A[3 * i + 2] * q.z;
					v = B[3 * i] * q.x + B[3 * i + 1] * q.y + B[3 * i + 2] * q.z;
				}
				if (u >= 0.0 && v >= 0.0 && u + v <= 1.0) {
					const double L = len(sub(P, O));
					if (dist == 0.0 || dist > L) { fp = P; hit = i; dist = L; }
				}
			}
		}
		double c[3] = {0.0, 1.0, 0.0};                    /* Green */
		if (hit >= 0) {
			if (dot(ld3(normals + 3 * hit), D) > 0.0) c[0] = c[1] = c[2] = 0.0;
			else {
				const v3 a = ld3(verts + 9 * hit), b = ld3(verts + 9 * hit + 3), cc = ld3(verts + 9 * hit + 6);
				const double ar = area3(a, b, cc);
				const double w1 = area3(fp, b, cc) / ar, w2 = area3(fp, a, cc) / ar, w3 = area3(fp, a, b) / ar;
				double f = 0.0;
				f += (w1 * vertex_rad[3 * hit]);
				f += (w2 * vertex_rad[3 * hit + 1]);
				f += (w3 * vertex_rad[3 * hit + 2]);
				c[0] = c[1] = c[2] = f;
			}
		}
		rgb[3 * k] = c[0]; rgb[3 * k + 1] = c[1]; rgb[3 * k + 2] = c[2];
	}
	free(A); free(B); free(inv);
}

/* ToneMapRadiosity, rasterizer.cpp:90-109: Green -> (0,255,0); else every channel = (channel)(255 · red/(red+1)) */
void rto_tone_map_radiosity(long n, const double *rgb, uint8_t *out) {
	for (long k = 0; k < n; ++k) {
		const double *c = rgb + 3 * k;
		if (c[0] == 0.0 && c[1] == 1.0 && c[2] == 0.0) { out[3 * k] = 0; out[3 * k + 1] = 255; out[3 * k + 2] = 0; continue; }
		const uint8_t g = (uint8_t)((double)255 * (c[0] / (c[0] + 1.0)));
		out[3 * k] = out[3 * k + 1] = out[3 * k + 2] = g;
	}
}
