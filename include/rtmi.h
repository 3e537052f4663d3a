/* rtmi.h — C ABI of librtmi.so: MI355X (gfx950) ray queries and radiosity for the
 * AryaGosuDev/RayTracer host driver.
 *
 * The reference has no FFI for this path; its seams are C++ members.  Each entry
 * point below names the reference interface it replaces (paths relative to the reference repo).
 * The host side (scene reader, OBJ loader, shader, PPM writer) stays the reference's own code and
 * calls these with *batches*: one call per wavefront of rays, not one call per ray.
 *
 * Conventions
 *   - plain C, opaque handles, int status: 0 = OK, negative = error; text via rtmi_last_error()
 *     (thread-local).  No exception crosses the boundary.
 *   - all geometry and rays are IEEE double, as in the reference (Render2/vec3.h:6-12).  The kernels
 *     evaluate the reference's Möller–Trumbore in double, in the reference's operation order, so hit
 *     ids and t/u/v are bit-identical to brute force over the input triangles with
 *     triangle_intersection (Render2/object.cpp:29-60).  Boxes are fp32 and conservatively padded.
 *   - "_dev" entry points take device pointers (resident in HBM) and a hipStream_t passed as void*;
 *     they are asynchronous on that stream.  The others take host pointers and are synchronous.
 *   - a scene lives on ONE device (the current HIP device at rtmi_scene_create).  Multi-GPU = one
 *     process (or one scene) per GPU; rays, image tiles and form-factor rows shard with no collective;
 *     only the radiosity solve exchanges data (the caller all-gathers B between sweeps).
 */
#ifndef RTMI_H
#define RTMI_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTMI_MISS 0xFFFFFFFFu
#define RTMI_INFINITY 1.0e20 /* Render2/base.h:56 — "no tmax" */

/* Render2/ray.h:23-32 (origin, unit direction); tmax <= 0 or >= 1e20 means 1e20.  reserved: 0 = a ray;
 * anything else = an inactive slot of a wavefront (answered with a miss / not occluded, nothing is traced) —
 * the generators below write it for pixels whose primary ray hit nothing.  64 bytes. */
typedef struct rtmi_ray {
	double o[3];
	double d[3];
	double tmax;
	double reserved;
} rtmi_ray;

/* What Scene::Cast leaves in HitInfo (Render2/ray_tracer.h:83-111) that is reproducible:
 * distance, barycentrics (float there too), the ORIGINAL triangle's flattened index and its
 * object.  tri == RTMI_MISS on a miss (then t = 0).  24 bytes. */
typedef struct rtmi_hit {
	double t;
	float u, v;
	uint32_t tri;
	uint32_t object;
} rtmi_hit;

/* Render2/rasterizer.cpp:11-68 (RasterDetails) as the kernels need it.  Filled by rtmi_camera_setup. */
typedef struct rtmi_camera {
	double eye[3];
	double O[3], dR[3], dU[3];
	uint32_t x_res, y_res;
} rtmi_camera;

typedef struct rtmi_scene rtmi_scene;

typedef struct rtmi_scene_info {
	uint64_t n_tris, n_nodes, n_leaves; /* nodes, leaves and depth of the binary hierarchy (the 4-wide one of the ray */
	uint32_t max_depth, max_leaf_size;  /* queries is its collapse; a leaf holds one triangle); device_bytes covers both */
	double build_seconds;
	uint64_t device_bytes;
	float scene_lo[3], scene_hi[3];
} rtmi_scene_info;

const char *rtmi_last_error(void);
int rtmi_device_count(void);

/* ---- the triangle test on its own ------------------------------------------------------------------
 * Replaces double triangle_intersection(orig, dir, planeNormal, v0, v1, v2, u, v)
 * (Render2/object.cpp:29-35, external linkage; planeNormal is unused there) on n independent pairs, host pointers:
 * o, d: n × 3; tri: n × 9 (v0, v1, v2); t[k] = 0.0 means miss; u, v are in/out and are written only
 * as far as the reference writes them before each early return. */
int rtmi_triangle_intersection(uint64_t n, const double *o, const double *d, const double *tri, double *t, double *u,
                               double *v);

/* ---- acceleration structure -------------------------------------------------------------------
 * Replaces Scene::BuildBSP (Render2/scene.cpp:463-479, 528-693).  tri_xyz: n_tris × 9 doubles
 * (triVert1, triVert2, triVert3) in Scene::sceneObjects / Object::triangles order — that order IS
 * the triangle id.  tri_object (may be NULL = all 0): index into sceneObjects.  Builds a binned-SAH
 * BVH on the host (one triangle per leaf), collapses it into the 4-wide nodes the ray queries walk, and uploads both
 * to the current device. */
rtmi_scene *rtmi_scene_create(const double *tri_xyz, const uint32_t *tri_object, uint64_t n_tris);
void rtmi_scene_destroy(rtmi_scene *);
int rtmi_scene_get_info(const rtmi_scene *, rtmi_scene_info *out);
/* A traversal wave that cannot finish (corrupt hierarchy, NaN rays) gives up after a bounded number of iterations per
 * ray instead of hanging the GPU, and raises the scene's error word; its launch leaves records unwritten.  Every later
 * entry point on the scene then fails with that message (sticky, like a device fault); rtmi_scene_check asks directly —
 * call it after synchronising the streams whose results you are about to trust.  No reference counterpart (the
 * reference's recursion cannot loop).  Launches of one scene may overlap on any number of streams: each stream handle that
 * has launched on the scene keeps its own 4-byte work-queue counter for the scene's life (counters come in blocks of 256; the
 * first launch on the 257th, 513th, ... distinct handle allocates a block — make it outside a graph capture). */
int rtmi_scene_check(const rtmi_scene *);

/* ---- closest hit --------------------------------------------------------------------------------
 * Replaces bool Scene::Cast(const Ray&, HitInfo&) (Render2/scene.cpp:159-207) and
 * Object::Intersect (Render2/object.cpp:170-184): strict minimum of the positive t below tmax;
 * equal t -> smallest triangle id. */
int rtmi_closest_hit(rtmi_scene *, const rtmi_ray *rays, uint64_t n, rtmi_hit *out);
int rtmi_closest_hit_dev(rtmi_scene *, const rtmi_ray *d_rays, uint64_t n, rtmi_hit *d_out, void *stream);

/* ---- primary rays generated on the device --------------------------------------------------------
 * rtmi_camera_setup replaces the RasterDetails constructor (Render2/rasterizer.cpp:40-56).
 * rtmi_primary_closest_hit_dev replaces the ray loop of Serial_Normal_Raster
 * (Render2/rasterizer.cpp:219-233): ray(i,j) = eye -> Unit(O + (j+sx)·dR − (i+sy)·dU), sx = sy = 0.5
 * in the reference.  The image is cut into tile_w × tile_h tiles numbered row-major; this call traces
 * tiles tile_first, tile_first + tile_stride, ... (tile_stride = number of GPUs) and writes
 * d_out[i·x_res + j] of a full x_res·y_res frame; other pixels are left untouched.
 * rtmi_primary_rays_dev writes the same rays out instead of tracing them (all pixels). */
int rtmi_camera_setup(const double eye[3], const double lookat[3], const double up[3], double vpdist,
                      const double x_win[2], const double y_win[2], uint32_t x_res, uint32_t y_res,
                      rtmi_camera *out);
int rtmi_primary_closest_hit_dev(rtmi_scene *, const rtmi_camera *cam, double sx, double sy, uint32_t tile_w,
                                 uint32_t tile_h, uint32_t tile_first, uint32_t tile_stride, rtmi_hit *d_out,
                                 void *stream);
int rtmi_primary_rays_dev(const rtmi_camera *cam, double sx, double sy, rtmi_ray *d_rays, void *stream);
/* Several samples per pixel in ONE launch (no counterpart in the reference, whose loop shoots the one ray through the
 * pixel centre): sample k shoots ray(i,j) with offsets (sx, sy)[k] through the pixels of tiles tile_first[k],
 * tile_first[k] + tile_stride, ... and writes its own full-size frame d_out[k] (frames may not overlap unless the
 * samples' tiles are disjoint).  Same rays and records as n_samples calls of rtmi_primary_closest_hit_dev, but one
 * work queue: the launches of a rank that holds 1/N of the tiles of N samples do not each end in their own tail.
 * `samples` is a HOST array (copied into the launch), n_samples <= RTMI_MAX_SAMPLES_PER_LAUNCH. */
#define RTMI_MAX_SAMPLES_PER_LAUNCH 16u
typedef struct rtmi_primary_sample {
	double sx, sy;        /* offsets inside the pixel, 0.5 = centre */
	uint32_t tile_first;  /* first tile of this sample */
	uint32_t reserved;    /* 0 */
	rtmi_hit *d_out;      /* device frame of x_res·y_res records */
} rtmi_primary_sample;
int rtmi_primary_closest_hit_samples_dev(rtmi_scene *, const rtmi_camera *cam, uint32_t tile_w, uint32_t tile_h,
                                         uint32_t tile_stride, const rtmi_primary_sample *samples, uint32_t n_samples,
                                         void *stream);
/* Diagnostic (no counterpart in the reference): the traversal work of the same frame of primary rays,
 * ADDED to d_counts[0] (node visits, two child boxes each), d_counts[1] (exact triangle tests) and
 * d_counts[2], d_counts[3] (lane slots the waves spent in the node loop / the triangle loop: 64 per
 * wave-iteration, so counts[0]/counts[2] is the lane occupancy of the node loop), d_counts[4], d_counts[5]
 * (lane-steps a lane sat out a node step because it held a leaf / had no ray).  d_counts: 6 × uint64.
 * bench.py prices the traversal kernel's roofline with it. */
int rtmi_primary_work_dev(rtmi_scene *, const rtmi_camera *cam, double sx, double sy, uint64_t *d_counts, void *stream);

/* ---- any hit --------------------------------------------------------------------------------------
 * Replaces Scene::Cast used as a predicate: shadows (Render2/shader.cpp:131, tmax = 1e20), ambient
 * occlusion (Render2/shader.cpp:226, tmax = radius), light visibility (Render2/radiosity.cpp:290).
 * occluded[k] = 1 iff some triangle has 0 < t < tmax. */
int rtmi_any_hit(rtmi_scene *, const rtmi_ray *rays, uint64_t n, uint8_t *occluded);
int rtmi_any_hit_dev(rtmi_scene *, const rtmi_ray *d_rays, uint64_t n, uint8_t *d_occluded, void *stream);

/* Host only, no GPU needed (tests): builds the hierarchies rtmi_scene_create would build for these triangles and checks
 * every invariant the kernels rely on — each child box of the 4-wide nodes, decoded as the kernels decode it, contains
 * the child's fp32 bounds; references in range; every triangle reachable exactly once; the stack bound holds on every
 * path.  stats = {violations, 4-wide nodes, stack entries needed, depth of the binary hierarchy}.  No reference
 * counterpart (the reference's BSP, scene.cpp:463-693, is replaced). */
int rtmi_debug_check_hierarchy(const double *tri_xyz, uint64_t n_tris, uint64_t stats[4]);

/* Test hook, no reference counterpart: n vector divisions (num3[3k..3k+2] / den[k], device pointers) by the library's
 * shared-reciprocal form (csrc/rtmi_device.hpp: div3, what every `Vec3 / double` of the kernels runs, vec3.h:46) and by the
 * compiler's three plain IEEE divisions; the tests require the two outputs to be the same bits. */
int rtmi_debug_div3_dev(const double *d_num3, const double *d_den, uint64_t n, double *d_shared3, double *d_plain3, void *stream);

/* Test hook, no reference counterpart: n unit vectors vec3[3k..3k+2] / |vec3[3k..3k+2]| by the form-factor kernel's bare sequence
 * (csrc/rtmi_device.hpp: unit_of_len2 — the compiler's square root and the three divisions without the range handling that does
 * nothing for operands away from the ends of the exponent range; a wave runs it only when all its lanes are in that range) and by
 * the plain root and divisions; the tests require the two outputs to be the same bits. */
int rtmi_debug_unit_dev(const double *d_vec3, uint64_t n, double *d_fast3, double *d_plain3, void *stream);

/* ---- wavefront: secondary rays generated on the device -----------------------------------------------
 * The batch producers between two ray-query calls, so that a frame's hit records never leave HBM:
 * d_rays[k] produced d_hits[k]; d_normals = n_tris × 3 doubles, the reference's triNormal per ORIGINAL triangle
 * (device pointer).  The hit point is the reference's plane formula (Render2/object.cpp:23).  Slots whose
 * hit is a miss are written as inactive rays.
 *   rtmi_shadow_rays_dev   Shader::generateNormalShadows (Render2/shader.cpp:116-127): origin P + 0.01·normal,
 *                          direction Unit(light_center − P); ask rtmi_any_hit_dev (shader.cpp:131)
 *   rtmi_ao_rays_dev       Shader::findOcclusionNew (Render2/shader.cpp:140-232): spp rays per hit, ray k·spp+s
 *                          from the square point of stream (seed, k, s) — the reference draws it from an unseeded
 *                          mt19937 —, origin P + 1e-4·dir, tmax = radius (occluded ⇔ Cast ∧ distance < radius, :226)
 *   rtmi_occlusion_fraction_dev   1 − occlusions/spp per hit (shader.cpp:231) from the any-hit bytes */
int rtmi_shadow_rays_dev(rtmi_scene *, const double *d_normals, const rtmi_ray *d_rays, const rtmi_hit *d_hits, uint64_t n,
                         const double light_center[3], rtmi_ray *d_out, void *stream);
int rtmi_ao_rays_dev(rtmi_scene *, const double *d_normals, const rtmi_ray *d_rays, const rtmi_hit *d_hits, uint64_t n,
                     uint32_t spp, double radius, uint64_t seed, rtmi_ray *d_out, void *stream);
int rtmi_occlusion_fraction_dev(const uint8_t *d_occluded, uint64_t n, uint32_t spp, double *d_out, void *stream);
/*   rtmi_shadow_occlusion_dev  rtmi_shadow_rays_dev + rtmi_any_hit_dev in one launch (Shader::generateNormalShadows,
 *                          Render2/shader.cpp:105-138): the shadow ray of each hit is built inside the any-hit kernel;
 *                          d_occluded[k] as the two calls would write it.  Misses give 0. */
int rtmi_shadow_occlusion_dev(rtmi_scene *, const double *d_normals, const rtmi_ray *d_rays, const rtmi_hit *d_hits, uint64_t n,
                              const double light_center[3], uint8_t *d_occluded, void *stream);
/*   rtmi_ao_occlusion_dev  the whole loop of Shader::findOcclusionNew (Render2/shader.cpp:172-229) in one launch:
 *                          d_occluded[k·spp + s] is byte for byte what rtmi_ao_rays_dev followed by rtmi_any_hit_dev
 *                          writes, but the spp rays per hit are built inside the any-hit kernel and never stored
 *                          (128 B of HBM traffic per ray less).  Misses give 0. */
int rtmi_ao_occlusion_dev(rtmi_scene *, const double *d_normals, const rtmi_ray *d_rays, const rtmi_hit *d_hits, uint64_t n,
                          uint32_t spp, double radius, uint64_t seed, uint8_t *d_occluded, void *stream);
/*   rtmi_soft_shadow_rays_dev    Shader::generateSoftShadows (Render2/shader.cpp:41-103; the shipped Shade has its call
 *                          commented out, :475,521): n_points (reference: 100 = a 25 × 4 grid, :55-71) rays per hit toward
 *                          the grid through the light centre, origin P + 0.05·direction; ask rtmi_closest_hit_dev, then
 *   rtmi_soft_shadow_factor_dev  1 − contributors/n_points per hit, a ray contributing iff it hits a triangle of ANOTHER
 *                          object than the shaded one (:93-100); the colour is multiplied by it */
int rtmi_soft_shadow_rays_dev(rtmi_scene *, const double *d_normals, const rtmi_ray *d_rays, const rtmi_hit *d_hits, uint64_t n,
                              const double light_center[3], uint32_t n_points, rtmi_ray *d_out, void *stream);
int rtmi_soft_shadow_factor_dev(const rtmi_hit *d_hits, const rtmi_hit *d_shadow_hits, uint64_t n, uint32_t n_points,
                                double *d_out, void *stream);
/*   rtmi_soft_shadow_dev  the same three steps in one launch + a reduction: the n_points rays per hit are built inside the
 *                          closest-hit kernel and never stored (the 3-step path moves 88 B per ray through HBM twice: 17 GB
 *                          for a 1920×1080 frame at n_points = 100); d_contrib (n · n_points bytes, workspace of the caller)
 *                          receives 1 per contributing ray, d_factor what rtmi_soft_shadow_factor_dev returns, bit for bit. */
int rtmi_soft_shadow_dev(rtmi_scene *, const double *d_normals, const rtmi_ray *d_rays, const rtmi_hit *d_hits, uint64_t n,
                         const double light_center[3], uint32_t n_points, uint8_t *d_contrib, double *d_factor, void *stream);

/* ---- frame: Scene::Trace + Shader::Shade as a wavefront -------------------------------------------------
 * Replaces the recursion Scene::Trace (Render2/scene.cpp:127-155) -> Shader::Shade (Render2/shader.cpp:395-645: the
 * branch the shipped main.cpp runs, scene.radiosity == NULL, and with RTMI_SHADE_RADIOSITY_BRANCH the other one,
 * :431-478) -> Scene::Trace ..., and the pixel loop + ToneMap of
 * Rasterizer::Serial_Normal_Raster (Render2/rasterizer.cpp:79-88, 219-233).  One generation of the ray tree is one
 * batch: closest hit, shadow ray of generateNormalShadows (:105-138), ambient-occlusion rays of findOcclusionNew
 * (:140-232), Phong / microfacet colour (:485-519), reflected (:527-537) and refracted (:542-641) children; the
 * returns (:614, :640, :643) are folded from the deepest generation up.  Colours are IEEE double in the reference's
 * operation order (pow/exp/sin/cos from the device library: a few ulp from the host's).  Work whose result the
 * reference never uses is skipped (AO of shadowed points or of materials with ambient 0, a reflected ray of weight 0).
 *
 * rtmi_material = Render2/ray_tracer.h:23-34 without emission/type; rtmi_light = what Shade reads of a light:
 * Center(GetBox(light)) and material->emission (shader.cpp:486-489).  Light 0 casts the shadows (shader.cpp:120).
 * object_materials[k] belongs to sceneObjects[k] = the tri_object ids the scene was created with.
 * tri_normals: n_tris × 3, the reference's triNormal per ORIGINAL triangle (host pointer, copied).
 * A shader holds a workspace (at most about RTMI_SHADE_WS_MB, default 8192, MiB; a frame is cut into chunks that
 * fit): calls on the same shader must not overlap. */
typedef struct rtmi_material {
	double diffuse[3], specular[3], ambient[3], reflectivity[3], translucency[3];
	double phong_exp, ref_index, roughness;
} rtmi_material;
typedef struct rtmi_light {
	double center[3];
	double emission[3];
} rtmi_light;
typedef struct rtmi_trace_params {
	uint32_t max_tree_depth; /* Scene::max_tree_depth, Render2/base.h:70: 1 */
	uint32_t ao_rays;        /* shader.cpp:481: 10; 0 = occlusion 1 (shader.cpp:142) */
	double ao_radius;        /* shader.cpp:481: 2.0 */
	uint64_t seed;           /* AO square points: stream (seed, ray-tree path, ray index, sample); the reference's are unseeded */
	uint32_t flags;          /* 0, or RTMI_SHADE_RADIOSITY_BRANCH */
	uint32_t reserved;       /* 0 */
} rtmi_trace_params;
/* RTMI_SHADE_RADIOSITY_BRANCH: Shade as it runs when scene.radiosity != NULL (Render2/shader.cpp:431-478), as shipped: the
 * radiosity lookup (trace_ray, :433) and the 100-ray occlusion (:434) are computed by the reference and used by nothing —
 * the lines that would use them are commented out (:457-465) — so the colour of a hit is the specular sum over the lights
 * (:467-472), returned (:477) before shadows, reflection and refraction; misses and the depth limit as in the other
 * branch.  The dead work is not done. */
#define RTMI_SHADE_RADIOSITY_BRANCH 1u
typedef struct rtmi_shader rtmi_shader;
rtmi_shader *rtmi_shader_create(rtmi_scene *, const double *tri_normals, const rtmi_material *object_materials,
                                uint32_t n_objects, const rtmi_light *lights, uint32_t n_lights);
void rtmi_shader_destroy(rtmi_shader *);
/* Color Scene::Trace(const Ray&) on n device-resident rays of the given generation (1 = from the eye,
 * rasterizer.cpp:217): d_rgb[3k..3k+2] before ToneMap.  Ray k is ray first_index + k of the AO stream.  Inactive
 * slots are left untouched. */
int rtmi_trace_dev(rtmi_shader *, const rtmi_ray *d_rays, uint64_t n, uint32_t generation, uint64_t first_index,
                   const rtmi_trace_params *params, double *d_rgb, void *stream);
/* The frame of Serial_Normal_Raster: this rank's tiles (as rtmi_primary_closest_hit_dev) of a full frame:
 * d_rgb[(i·x_res + j)·3 ..] doubles and/or d_pixels[(i·x_res + j)·3 ..] = ToneMap bytes (either may be NULL). */
int rtmi_render_frame_dev(rtmi_shader *, const rtmi_camera *cam, double sx, double sy, uint32_t tile_w, uint32_t tile_h,
                          uint32_t tile_first, uint32_t tile_stride, const rtmi_trace_params *params, double *d_rgb,
                          uint8_t *d_pixels, void *stream);
/* The other two pixel loops of the reference's Rasterizer, same frame conventions:
 *   RTMI_RASTER_ANTI_ALIASING    Rasterizer::Anti_Aliasing (Render2/rasterizer.cpp:297-349): 4 rays per pixel at
 *                                (j + {.25,.75}, i + {.25,.75}), colours summed in the reference's loop order, / 4.0
 *   RTMI_RASTER_DEPTH_OF_FIELD   Rasterizer::Depth_Of_Field_Effect (:351-406): lens_dim² rays per pixel from the lens
 *                                grid eye + xLens·XGradation + yLens·YGradation toward the view plane re-aimed at lookat
 *   RTMI_RASTER_NORMAL           the loop of rtmi_render_frame_dev (sample position sx, sy)
 * rtmi_raster_setup fills the RasterDetails fields these loops read (:40-68: R, U, XGradation = Lwidth·R/7,
 * YGradation = Lheight·U/7, DistLensDim 7).  Sample s of pixel p is ray p·S + s of the AO stream. */
#define RTMI_RASTER_NORMAL 0u
#define RTMI_RASTER_ANTI_ALIASING 1u
#define RTMI_RASTER_DEPTH_OF_FIELD 2u
typedef struct rtmi_raster {
	uint32_t mode, lens_dim;
	double sx, sy;
	double lookat[3], R[3], U[3], x_gradation[3], y_gradation[3];
	double vpdist, xmin, ymax;
} rtmi_raster;
int rtmi_raster_setup(uint32_t mode, const double eye[3], const double lookat[3], const double up[3], double vpdist,
                      const double x_win[2], const double y_win[2], rtmi_raster *out);
int rtmi_render_frame_sampled_dev(rtmi_shader *, const rtmi_camera *cam, const rtmi_raster *raster, uint32_t tile_w,
                                  uint32_t tile_h, uint32_t tile_first, uint32_t tile_stride,
                                  const rtmi_trace_params *params, double *d_rgb, uint8_t *d_pixels, void *stream);
/* ToneMap (Render2/rasterizer.cpp:79-88) on n colours -> 3n bytes */
int rtmi_tone_map_dev(const double *d_rgb, uint64_t n, uint8_t *d_pixels, void *stream);

/* ---- scene files (host only, no GPU) ------------------------------------------------------------------
 * Replaces bool Scene::BuildScene(file, fileObj, camera) (Render2/scene.cpp:209-460) with ParamReader
 * (Render2/params.cpp): reads <scenes_dir>/<stem>.sdf and the <object name>.obj files it names, and hands out exactly
 * the arrays the entry points above take: triangles in sceneObjects / Object::triangles order (quads split as
 * params.cpp:214-241), triNormal per triangle (params.cpp:188-235), one material per object, the lights whose emission
 * is not black (scene.cpp:276) as Center(GetBox(light)) + emission, and the camera as rtmi_camera_setup's arguments
 * (cam14 = eye, lookat, up, vpdist, x_win, y_win).  Any pointer of rtmi_scene_file_get may be NULL. */
typedef struct rtmi_scene_file rtmi_scene_file;
rtmi_scene_file *rtmi_scene_file_load(const char *scenes_dir, const char *stem);
/* The same reader without the limits that keep the reference's larger assets from loading (Render2/scene.cpp:211: 512-byte
 * lines; :400-405: the first line must be `o <object name>`; Render2/params.cpp:214: triangles and quads only):
 * RTMI_SCENE_LENIENT takes lines of any length, any `o` / `g` name (or none), relative (negative) indices, and faces with any
 * number of corners — v1 v2 v3 v4 v5 ... becomes (v1,v2,v3), (v3,v4,v1), (v4,v5,v1), ...: the reference's quad split, carried on.
 * A missing .obj is an error in this mode.  flags = 0 is rtmi_scene_file_load. */
#define RTMI_SCENE_LENIENT 1u
rtmi_scene_file *rtmi_scene_file_load_ex(const char *scenes_dir, const char *stem, uint32_t flags);
void rtmi_scene_file_free(rtmi_scene_file *);
int rtmi_scene_file_info(const rtmi_scene_file *, uint64_t *n_tris, uint32_t *n_objects, uint32_t *n_lights);
int rtmi_scene_file_get(const rtmi_scene_file *, double *tri_xyz, uint32_t *tri_object, double *tri_normals,
                        rtmi_material *materials, rtmi_light *lights, double cam14[14], uint32_t res[2]);

/* ---- radiosity: form factors ------------------------------------------------------------------------
 * The patches are the scene's triangles (elements = the leaves every visibility ray is tested
 * against, Render2/radiosity.cpp:137-163), so patch i = triangle i of the scene.  verts: n × 9 doubles,
 * normals: n × 3 doubles = the reference's triNormal per element (host pointers, copied). */
typedef struct rtmi_patches rtmi_patches;
rtmi_patches *rtmi_patches_create(rtmi_scene *, const double *verts /* the array the scene was built from */,
                                  const double *normals);
void rtmi_patches_destroy(rtmi_patches *);

/* Replaces bool Radiosity::isVisibleXiToXj (Render2/radiosity.cpp:231-262) on m injected point pairs
 * (host pointers): vis[k] = closest hit from xi+1e-5·d lands within 1e-5 of xj  and  d·n_ej < 0. */
int rtmi_visible_points(rtmi_patches *, uint64_t m, const int32_t *ej, const double *xi, const double *xj,
                        uint8_t *vis);

#define RTMI_FF_LIGHT_MEAN 1u /* average all light samples instead of mirroring Render2/radiosity.cpp:276 */

/* Replaces Radiosity::findFormFactor's two fill loops (Render2/radiosity.cpp:379-403) with
 * findFormFactorTerm (:305-333) and findFormFactorTermWithLight (:264-303).  Sample points come from
 * the counter-based generator documented in DESIGN.md (stream = seed, i, j, sample).
 * n = n_patches + 1 (the light is patch n-1, its row is zero) if light_center != NULL, else n_patches.
 * Writes rows [row0, row0+n_rows) of the n × n matrix as fp32: d_F[(r-row0)·ld + c].  Rows shard
 * across GPUs with no communication.  ld >= n, a multiple of 4 (16-byte rows); choose a multiple of 32 (rows that start on
 * 128-byte lines): the sweeps that read the matrix back run 8 % faster on it (64 745 columns: 2.93 -> 2.69 ms per sweep).
 * Two kernels compute the same matrix bit for bit: one follows the two patches' paths through the hierarchy
 * (fast in rooms), one descends from the root (fast in clutter).  The first call on a patch set that covers at least 256 patch
 * rows of at least 1024 patches times both on its first rows (up to 1024) and keeps the faster for that patch set — that one call
 * waits for the measurement; RTMI_FF_KERNEL=path|flat in the environment forces one.  A call whose stream is being captured
 * into a graph measures nothing (it takes what an earlier call settled, else the path kernel); the per-patch-set tables of the
 * path kernel are built (hipMalloc, copies) by the FIRST call that uses it, so make one call outside the capture first.
 * rtmi_debug_form_factor_kernel: 0 = not measured yet, 1 = paths, 2 = root (diagnostic; no reference counterpart).
 * rtmi_debug_form_factor_planes: 1 = the path kernel's tables carry the "sibling lies behind this patch's plane" flags, 0 = they
 * do not (caller normals that are not the triangles' geometric ones, |coordinates| beyond ~1 400, RTMI_FF_PLANES=0), -1 = the
 * tables are not built yet (diagnostic; no reference counterpart). */
int rtmi_debug_form_factor_kernel(rtmi_patches *);
int rtmi_debug_form_factor_planes(rtmi_patches *);
int rtmi_form_factors_dev(rtmi_patches *, const double *light_center, uint32_t samples, uint32_t light_samples,
                          uint64_t seed, uint32_t flags, uint32_t row0, uint32_t n_rows, float *d_F, uint64_t ld,
                          void *stream);

/* ---- radiosity: solve ----------------------------------------------------------------------------------
 * Replaces Radiosity::progressiveRefinement + checkIfConverged (Render2/radiosity.cpp:483-563,
 * 469-481): both iterate to the solution of (I − ρF)·B = E; here by Jacobi sweeps
 *     B_new[i][c] = E[i][c] + rho[c] · Σ_j F[i][j] · B_old[j][c]      c = R,G,B
 * F fp32 (streamed from HBM once per sweep), B/E fp32 [n][3], accumulation in double.
 * One call = one sweep over rows [row0, row0+n_rows): reads d_B_old (all n), writes
 * d_B_new[row0..row0+n_rows) (same [n][3] indexing) and ADDS to d_norms[0] += Σ(B_new−B_old)²,
 * d_norms[1] += Σ(B_new−E)² (the gathered part ρ·F·B: an emitter's own power — the light's 68 500 —
 * must not set the scale), d_norms[2] = max(d_norms[2], max|B_new−B_old|) over its rows
 * (the caller zeroes d_norms; [2] is the reference's stop quantity, :478).
 * Multi-GPU: rows shard; the caller all-gathers d_B_new between sweeps (RCCL). */
int rtmi_jacobi_sweep_dev(const float *d_F, uint64_t ld, uint32_t n, uint32_t row0, uint32_t n_rows,
                          const double rho[3], const float *d_E, const float *d_B_old, float *d_B_new,
                          double *d_norms, void *stream);

typedef struct rtmi_solve_stats {
	uint32_t sweeps;
	double rel_change; /* ‖B_k − B_{k-1}‖₂ / ‖B_k − E‖₂ at exit */
	double max_change; /* max_i |B_k − B_{k-1}| at exit = max residual of B_{k-1} */
	double seconds;    /* device time of the sweeps (HIP events) */
	double bytes_per_sweep;
	uint32_t accelerated_sweeps; /* how many of `sweeps` were Chebyshev steps (0: Jacobi throughout) */
	double spectral_radius;      /* the estimate of rho(rho·F) the Chebyshev steps used (0: none was made) */
} rtmi_solve_stats;

/* exchange hook: called after every sweep with the stream the sweep ran on; must make
 * d_B (all n rows) hold every rank's rows and sum d_norms[0..1] / max d_norms[2] across ranks
 * before returning (stream-ordered).  NULL = single GPU.
 * It is also called ONCE BEFORE THE FIRST SWEEP (d_B = B_0 = E, the same on every rank; d_norms = {0, 0, status}): the
 * maximum it takes of d_norms[2] is how the ranks agree that every one of them got through its set-up — a rank that did
 * not (hipMalloc failed) puts +inf there and EVERY rank returns -1 instead of waiting in a collective for a peer that left.
 * A rank whose sweep fails later keeps calling the hook with +inf in d_norms[2] until the next check, where all return -1. */
typedef int (*rtmi_exchange_fn)(void *user, float *d_B, double *d_norms, void *stream);

/* Whole solve on device-resident F (rows [row0,row0+n_rows) local).  d_B [n][3] receives the
 * result (every rank gets all rows when exchange != NULL).  Stops when rel_change <= tol_rel
 * (checked every check_every sweeps) or after max_sweeps.
 * The first window (check_every sweeps, at most 3) is Jacobi; from the first check on the sweeps are Chebyshev steps on the same
 * kernel (B_{k+1} = w·J(B_k) + (1 − w)·B_{k-1}; the spectral radius of rho·F is estimated from the Jacobi sweeps' change
 * norms): the same fixed point in about half the sweeps.  RTMI_SOLVER=jacobi in the environment keeps Jacobi throughout.
 * tol_rel = 0: exactly max_sweeps Jacobi sweeps and no host read of the norms (rel_change / max_change come back NaN).
 * The call allocates its scratch (hipMalloc) and synchronises the stream at every check and at the end: it cannot be
 * captured into a graph (rtmi_radiosity_solve_noalloc_dev with tol_rel = 0 can). */
int rtmi_radiosity_solve_dev(const float *d_F, uint64_t ld, uint32_t n, uint32_t row0, uint32_t n_rows,
                             const double rho[3], const float *d_E, float *d_B, float tol_rel, uint32_t max_sweeps,
                             uint32_t check_every, rtmi_exchange_fn exchange, void *user, rtmi_solve_stats *stats,
                             void *stream);

/* The same solve across GPUs, the exchange INSIDE the library (RCCL over xGMI): for hosts that are not Python — the
 * reference's driver is a C++ program (Render2/main.cpp:6-49).  No reference counterpart (its solver is one thread).
 * One process (or thread) per GPU; F is row-sharded in equal blocks: with blk = ceil(n / world), rank r owns rows
 * [min(n, r·blk), min(n, (r+1)·blk)) — pass them as row0 / n_rows, with d_F pointing at the rank's first row.
 *   rtmi_comm_unique_id   rank 0 draws an id (ncclGetUniqueId) and hands it to the other ranks by its own means
 *   rtmi_comm_create      every rank, on its own GPU (the current device), with the same id: collective
 *   rtmi_radiosity_solve_sharded_dev   collective: every rank calls it with the same n, tolerances and check_every.
 *     Per sweep: the sweep kernel on this rank's rows, then an in-place all-gather of the ranks' blocks of B — both only
 *     enqueued on `stream`; every check_every sweeps two small all-reduces of the norms and ONE host wait.  d_B [n][3]
 *     receives all rows on every rank; stats->seconds is host wall time (sweeps, exchange and checks).  It is the same loop
 *     as rtmi_radiosity_solve_dev's (same sweeps, same Chebyshev steps, same B bit for bit).  Before the first sweep the
 *     ranks all-reduce a status word: a rank whose set-up failed makes every rank return -1; a rank whose sweep launch fails
 *     later still enters the window's collectives and poisons the norms, so all ranks return -1 at the next check; an RCCL
 *     error aborts the communicator (ncclCommAbort) — it must then be destroyed and made again.
 * RCCL is loaded at run time (dlopen librccl.so.1); without it these three fail with a message, nothing else needs it. */
#define RTMI_COMM_ID_BYTES 128
typedef struct rtmi_comm rtmi_comm;
int rtmi_comm_unique_id(uint8_t id[RTMI_COMM_ID_BYTES]);
rtmi_comm *rtmi_comm_create(const uint8_t id[RTMI_COMM_ID_BYTES], int rank, int world);
void rtmi_comm_destroy(rtmi_comm *);
int rtmi_radiosity_solve_sharded_dev(rtmi_comm *, const float *d_F, uint64_t ld, uint32_t n, uint32_t row0, uint32_t n_rows,
                                     const double rho[3], const float *d_E, float *d_B, float tol_rel, uint32_t max_sweeps,
                                     uint32_t check_every, rtmi_solve_stats *stats, void *stream);

/* The same solve for callers that cannot allocate.  NO COUNTERPART IN THE REFERENCE: it exists for hosts that must not
 * call hipMalloc — code under HIP graph capture, tight serving loops, harnesses that forbid allocation.  It is
 * rtmi_radiosity_solve_dev with exchange = NULL over all n rows (same kernels, same results bit for bit), but the solver's
 * scratch — the second B vector of the Jacobi ping-pong (12·n bytes) and the norm windows — lives in a static device
 * array inside the library (about 1.6 MiB per GPU, held until the process ends) instead of being hipMalloc'ed per call,
 * and no HIP events are created (stats->seconds is host wall time around the loop).  It calls no hipMalloc, hipFree or
 * other allocating routine.  Limits: n <= RTMI_NOALLOC_MAX_N and check_every <= RTMI_NOALLOC_MAX_CHECK_EVERY, otherwise -1
 * with a message.  Every call on one GPU shares that GPU's array: calls that overlap on one GPU (two host threads, two
 * streams) race — serialise them.  What must start at zero (the norm windows) is cleared on the stream by each call.
 * With a tolerance the call reads the norms on the host at every check and waits for the stream at the end, as
 * rtmi_radiosity_solve_dev does: such a call is REFUSED (-1) while `stream` is being captured.  With tol_rel = 0 it
 * enqueues exactly max_sweeps Jacobi sweeps and the copy into d_B and reads nothing back; on a capturing stream it does not
 * synchronise either — that form can be recorded into a HIP graph and replayed (tests/test_zz_no_alloc.py does). */
#define RTMI_NOALLOC_MAX_N 131072u
#define RTMI_NOALLOC_MAX_CHECK_EVERY 64u
int rtmi_radiosity_solve_noalloc_dev(const float *d_F, uint64_t ld, uint32_t n, const double rho[3], const float *d_E,
                                     float *d_B, float tol_rel, uint32_t max_sweeps, uint32_t check_every,
                                     rtmi_solve_stats *stats, void *stream);

/* ---- radiosity: display ------------------------------------------------------------------------------
 * Replaces the pixel loop of Rasterizer::Radiosity_Raster (Render2/rasterizer.cpp:408-470):
 * I(i,j) = ToneMapRadiosity(radiosityHelper->trace_ray(ray, elements)), with
 *   Radiosity_Helper::trace_ray (Render2/radiosity_helper.cpp:318-365): every element is intersected on the whole LINE
 *     of the ray (t of either sign), inside ⇔ u ≥ 0 ∧ v ≥ 0 ∧ u+v ≤ 1 from returnUVofTriangle's pseudo-inverse
 *     (Render2/util.cpp:126-147, mat3x2.h:187-283, computed on the host at rtmi_patches_create), nearest |P − origin|
 *     wins; no element -> Green (0,1,0), back face -> Black;
 *   the barycentric half of returnQuadTreeColor (Render2/radiosity_helper.h:235-267): Σ_k Area-weight_k · radiosity at
 *     triVert_k.  d_vertex_radiosity: n_patches × 3 doubles (device) — the value returnQuadTreeColor attaches to each
 *     vertex (its shared-vertex averages over QuadTreeNode::nextAdj are the host's; the adapter reads them off the
 *     reference by evaluating it at the vertices); for flat shading pass B[i] three times;
 *   ToneMapRadiosity (Render2/rasterizer.cpp:90-109).
 * The query is the reference's O(elements) loop per ray, tiled through LDS; rays and frame conventions as above. */
int rtmi_radiosity_trace_dev(rtmi_patches *, const double *d_vertex_radiosity, const rtmi_ray *d_rays, uint64_t n,
                             double *d_rgb, void *stream);
int rtmi_radiosity_raster_dev(rtmi_patches *, const double *d_vertex_radiosity, const rtmi_camera *cam, double sx, double sy,
                              uint32_t tile_w, uint32_t tile_h, uint32_t tile_first, uint32_t tile_stride, double *d_rgb,
                              uint8_t *d_pixels, void *stream);
int rtmi_tone_map_radiosity_dev(const double *d_rgb, uint64_t n, uint8_t *d_pixels, void *stream);

/* The first half of returnQuadTreeColor (Render2/radiosity_helper.h:156-213) for hosts without the reference's QuadTreeNode
 * graph: the radiosity it attaches to each vertex of each element — over the elements a that share the vertex p with element e
 * (QuadTreeNode::nextAdj), the mean of
 *     (1 − |c_e − p| / full)·B_e + (1 − |p − c_a| / full)·B_a,   full = |c_e − p| + |p − c_a|,
 * c = centerOfTriangle (:115-142: the crossing of two medians through calculatePseudoInverse), and B_e itself where nobody
 * shares the vertex (:255-257).  The reference's own adjacency build is switched off (Render2/radiosity.cpp:167-199), so as
 * shipped every vertex gets B_e (flat shading); the adjacency used here is "the same three doubles" — what an OBJ's shared
 * vertex index gives — summed in element order.  Host code, host pointers: verts [n][9] (triVert1..3), B [n] (one channel),
 * out [n][3] = what rtmi_radiosity_raster_dev / _trace_dev take as vertex radiosity.  Pinned against the compiled reference's
 * own returnQuadTreeColor with nextAdj filled with that adjacency (tests/golden/g10_vertex_values.npz). */
int rtmi_radiosity_vertex_values(const double *verts, uint32_t n, const double *B, double *out);

#ifdef __cplusplus
}
#endif
#endif
