/* TEST INFRASTRUCTURE — not part of the product.
 *
 * rt_oracle: a plain-C, all-double CPU restatement of the reference's hot path
 * (AryaGosuDev/RayTracer, Render2/), every function citing the file:line it
 * follows.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load liboracle.so; the product (raytracer_amd/, librtmi.so) never does.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this restatement
 * against vectors produced by the compiled reference itself (oracle/_ref, see
 * oracle/gen_golden.py): triangle_intersection, RasterDetails, Scene::Cast,
 * isVisibleXiToXj, castFromElementToElement, progressiveRefinement.
 *
 * Arithmetic is IEEE double in the reference's own operation order; build with
 * -ffp-contract=off so no fused multiply-add changes a rounding.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTO_INFINITY 1.0e20           /* base.h:56 */
#define RTO_EPSILON 1.0e-5            /* base.h:57 */
#define RTO_PI 3.14159265358979       /* base.h:52 */
#define RTO_REFLECTIVITY 0.70         /* base.h:42 */
#define RTO_LIGHT_POWER (1370.0 * 50.0) /* base.h:44 */

/* object.cpp:29-60.  Returns t, 0.0 = miss.  *u,*v are written only as far as the
 * reference writes them (callers preset them). */
double rto_mt(const double o[3], const double d[3], const double v0[3], const double v1[3], const double v2[3],
              double *u, double *v);
void rto_mt_batch(long n, const double *o, const double *d, const double *v0, const double *v1, const double *v2,
                  double *t, double *u, double *v);

/* Closest hit = strict minimum of the positive t over tris[0..n_tris) in order, first
 * wins on equal t, t must be below tmax (reference: Infinity = 1e20).
 * object.cpp:143-168 (intersectTreeTraversalNoAccel) + scene.cpp:190-206.
 * tris: n_tris × 9 doubles (v0,v1,v2).  tmax may be NULL (= 1e20 for every ray).
 * tri[k] = -1 on a miss (then t=u=v=0). */
void rto_closest_hit(long n_tris, const double *tris, long n_rays, const double *o, const double *d,
                     const double *tmax, int32_t *tri, double *t, double *u, double *v);

/* Any hit: ∃ triangle with 0 < t < tmax.  shader.cpp:131 (tmax = 1e20), shader.cpp:226 (tmax = radius). */
void rto_any_hit(long n_tris, const double *tris, long n_rays, const double *o, const double *d, const double *tmax,
                 uint8_t *occluded);

/* rasterizer.cpp:40-56: cam14 = eye[3] lookat[3] up[3] vpdist xwin[2] ywin[2]; out18 = G,U,R,O,dR,dU */
void rto_raster_details(const double *cam14, uint32_t x_res, uint32_t y_res, double *out18);
/* rasterizer.cpp:230: dir(i,j) = Unit(O + (j+.5)dR − (i+.5)dU); pix = n × (row i, col j) */
void rto_primary_dirs(const double *rd18, long n, const uint32_t *pix, double *dir);

/* rasterizer.cpp:219-233 (mode 0, S = 1), :297-349 (mode 1 anti-aliasing, S = 4), :351-406 (mode 2 depth of field,
 * S = 49): the S generation-1 rays of each listed pixel in the reference's loop order; o, d: n·S × 3 */
void rto_raster_rays(int mode, const double *cam14, uint32_t x_res, uint32_t y_res, long n, const uint32_t *pix, double *o,
                     double *d);

/* object.cpp:23: point = o + (d·((v1−o)·n)) / (n·d) */
void rto_hit_point(const double o[3], const double d[3], const double v1[3], const double nrm[3], double p[3]);

/* shader.cpp:116-127: shadow ray origin P + 0.01·n, direction Unit(C − P) */
void rto_shadow_rays(long n, const double *P, const double *N, const double C[3], double *o, double *d);

/* shader.cpp:41-103 (generateSoftShadows, uncalled by the shipped Shade): n_points (reference 100 -> 25 × 4 grid) rays per
 * point toward the grid on the plane z = C.z, origin P + 0.05·dir.  The caller asks for the CLOSEST hit and counts the
 * rays whose hit belongs to another object than the shaded point's (:93); factor = 1 − count/n_points. */
void rto_soft_shadow_rays(long n, const double *P, const double C[3], uint32_t n_points, double *o, double *d);

/* shader.cpp:140-232 (findOcclusionNew): spp AO rays per point; ray k·spp+s uses the square point
 * rto_ao_square_point(seed, k, s) where the reference draws from an unseeded mt19937.  Origin P + 1e-4·dir;
 * the caller asks "hit closer than R" (shader.cpp:226). */
void rto_ao_square_point(uint64_t seed, uint64_t ray, uint32_t sample, double ab[2]);
void rto_ao_rays(long n, const double *P, const double *N, double R, uint32_t spp, uint64_t seed, double *o, double *d);
void rto_ao_rays_points(long m, const double *P, const double *N, double R, const double *ab, double *o, double *d);

/* ---- radiosity ---------------------------------------------------------- */
/* patches: n × 9 doubles (v1,v2,v3) + normals n × 3 (the reference's triNormal).  The
 * patches are also the occluders (elements = the BSP leaves every ray is tested against). */

/* vec3.h:116-118 */
double rto_area(const double v1[3], const double v2[3], const double v3[3]);

/* radiosity.cpp:231-262 + 202-229 on injected points */
void rto_visible(long n, const double *verts, const double *normals, long m, const int32_t *ei, const int32_t *ej,
                 const double *xi, const double *xj, uint8_t *vis);

/* radiosity.cpp:279-290: ray xi + 1e-5·n_i → Unit(C − xi); blocked ⇔ any triangle has t > 0 (t < 1e20) */
void rto_light_blocked(long n, const double *verts, const double *normals, long m, const int32_t *ei,
                       const double *xi, const double C[3], uint8_t *blocked);

/* counter-based sample generator shared (by construction, not by code) with the HIP kernels:
 * uniform point on patch e for stream (seed, a, b, k, which).  util.cpp:113-115 for the
 * barycentric combination; the reference's rejection loop (util.cpp:104-110) is replaced by the
 * reflection (u,v) -> (1-u,1-v), which samples the same uniform distribution. */
uint64_t rto_hash(uint64_t seed, uint32_t a, uint32_t b, uint32_t k, uint32_t which);
void rto_sample_point(const double *verts9, uint64_t seed, uint32_t a, uint32_t b, uint32_t k, uint32_t which,
                      double p[3]);

/* radiosity.cpp:305-333 with samples from rto_sample_point(seed, i, j, k, 0|1).
 * F: n_rows × ld doubles, rows [row0,row0+n_rows), columns [0,n); F[i][i] = 0 (radiosity.cpp:382). */
void rto_form_factors(long n, const double *verts, const double *normals, uint32_t samples, uint64_t seed,
                      long row0, long n_rows, long ld, double *F);
/* same formula on injected points: xi,xj = S points each for ONE pair (i,j) */
double rto_ff_term_points(long n, const double *verts, const double *normals, int32_t i, int32_t j, uint32_t S,
                          const double *xi, const double *xj);

/* radiosity.cpp:264-303 with samples from rto_sample_point(seed, i, 0xFFFFFFFF, k, 0).
 * mirror_bug != 0: the accumulator is reset inside the loop (:276), so only the last sample
 * counts and is then divided by S — what the reference computes.  0: the mean over S. */
void rto_light_terms(long n, const double *verts, const double *normals, const double C[3], uint32_t S,
                     uint64_t seed, int mirror_bug, long row0, long n_rows, double *f);
double rto_light_term_points(long n, const double *verts, const double *normals, const double C[3], int32_t i,
                             uint32_t S, const double *xi, int mirror_bug);

/* radiosity.cpp:483-563 + 469-481 + util.cpp:84-94: progressive-refinement shooting on an
 * (n×n, row-major, incl. the light as patch n−1) form-factor matrix.  area: n−1 element areas.
 * The reference reads tempdBArea[n−1] uninitialised (:530-533); it is taken as 0 here.
 * Returns the number of shots, or −1 if max_shots was reached. */
long rto_progressive_refinement(long n, const double *F, const double *area, double rho, double power,
                                long max_shots, double *B);

/* The fixed point itself: solve (I − ρF)·B = E by Gaussian elimination with partial pivoting
 * (E = 0 except E[n−1] = power; light row = identity).  K as radiosity.cpp:413-428. */
int rto_direct_solve(long n, const double *F, double rho, double power, double *B);

/* Jacobi sweeps B ← E + ρ∘(F·B) in double, 3 right-hand sides (RGB): rho[3], E n×3, B n×3.
 * Stops when ‖B_new − B_old‖₂ ≤ tol_rel·‖B_new‖₂ ; returns sweeps done. */
long rto_jacobi_rgb(long n, const float *F, const double rho[3], const double *E, double tol_rel, long max_sweeps,
                    double *B);

/* ---- CPU baseline: same closest-hit contract through a BVH (our own, binned SAH) ---- */
typedef struct rto_bvh rto_bvh;
/* rto_form_factors with every closest hit taken through the hierarchy (same records; for the 64 K-patch rows) */
void rto_form_factors_bvh(const rto_bvh *, long n, const double *verts, const double *normals, uint32_t samples, uint64_t seed,
                          long row0, long n_rows, long ld, double *F);
rto_bvh *rto_bvh_build(long n_tris, const double *tris);
void rto_bvh_free(rto_bvh *);
/* identical results to rto_closest_hit (tests pin this); returns seconds, uses OpenMP threads */
double rto_bvh_closest_hit(const rto_bvh *, long n_rays, const double *o, const double *d, const double *tmax,
                           int32_t *tri, double *t, double *u, double *v);
int rto_num_threads(void);

/* ---- Scene::Trace + Shader::Shade (scene.cpp:127-155, shader.cpp:395-645; the scene.radiosity == NULL branch) ----
 * mats18: per object {diffuse, specular, ambient, reflectivity, translucency}[3] + Phong_exp, ref_index, roughness;
 * lights6: per light {Center(GetBox(light)), emission}; normals = triNormal per triangle; bvh may be NULL (in-order
 * scan).  Ray k has index first_index + k; the AO square points of the node reached from it along `path` (1 = the ray
 * itself, 2p = reflected / totally reflected child of p, 2p+1 = refracted child) come from
 * rto_ao_square_point(mix64(seed ^ path), index, sample) where the reference draws from an unseeded mt19937.
 * rgb: n_rays × 3 doubles before ToneMap. */
void rto_trace(long n_tris, const double *tris, const double *normals, const uint32_t *tri_object, const double *mats18,
               const double *lights6, int n_lights, const rto_bvh *bvh, uint32_t max_tree_depth, uint32_t ao_rays,
               double ao_radius, uint64_t seed, long n_rays, const double *o, const double *d, uint32_t generation,
               uint64_t first_index, double *rgb);
/* the same with flags: RTO_SHADE_RADIOSITY = Shader::Shade's radiosity branch (shader.cpp:431-478, scene.radiosity != NULL)
 * as shipped: the colour is the specular sum over the lights, nothing else */
#define RTO_SHADE_RADIOSITY 1u
void rto_trace_flags(long n_tris, const double *tris, const double *normals, const uint32_t *tri_object, const double *mats18,
                     const double *lights6, int n_lights, const rto_bvh *bvh, uint32_t max_tree_depth, uint32_t ao_rays,
                     double ao_radius, uint64_t seed, uint32_t flags, long n_rays, const double *o, const double *d,
                     uint32_t generation, uint64_t first_index, double *rgb);
/* rasterizer.cpp:79-88 */
void rto_tone_map(long n, const double *rgb, uint8_t *out);

/* ---- radiosity display path (Rasterizer::Radiosity_Raster, rasterizer.cpp:408-470) ----
 * rto_pinv: calculatePseudoInverse (mat3x2.h:187-283) as returnUVofTriangle uses it (util.cpp:126-147).
 * rto_radiosity_trace: Radiosity_Helper::trace_ray (radiosity_helper.cpp:318-365) + the barycentric half of
 * returnQuadTreeColor (radiosity_helper.h:235-267) with per-vertex radiosities injected (vertex_rad n × 3).
 * rto_tone_map_radiosity: ToneMapRadiosity (rasterizer.cpp:90-109). */
int rto_pinv(const double tri9[9], double A[3], double B[3]);
void rto_radiosity_trace(long n_el, const double *verts, const double *normals, const double *vertex_rad, long n_rays,
                         const double *o, const double *d, double *rgb);
void rto_tone_map_radiosity(long n, const double *rgb, uint8_t *out);

#ifdef __cplusplus
}
#endif
#endif
