// TEST INFRASTRUCTURE — not part of the product.  Only tests/, the golden-vector
// generator (oracle/gen_golden.py) and bench.py's cpu_baseline leg may load the
// library this file is linked into (oracle/_ref/libref_render2.so).
//
// This is OUR driver around the *compiled reference*: oracle/Makefile compiles the
// reference's 14 project sources where they lie (/root/reference/Render2/*.cpp,
// main.cpp and the dead abvh.cpp excluded) and links them with this file.  Every
// entry point below calls the reference's own functions and only marshals flat
// arrays in and out:
//   triangle_intersection            Render2/object.cpp:29-60
//   Object::Intersect / Scene::Cast  Render2/object.cpp:170-184, scene.cpp:159-207
//   Scene::BuildScene / BuildBSP     Render2/scene.cpp:209-460, 463-479
//   Radiosity::*                     Render2/radiosity.cpp (phase by phase, public members)
// Nothing here runs on the GPU box unless bench.py times it as the CPU baseline.
#include "ray_tracer.h"
#include "util.h"
#include "radiosity_helper.h"
#include "ref_scene.h"

#include <chrono>
#include <cstdint>
#include <cstring>
#include <unistd.h>

// main.cpp:4 defines this static; main.cpp is not linked.
int BSP_Node::numOfTriangle = 1;

// const-reference forwarding overloads declared in ref_compat.h (see there).
bool doesPointLieOnLine(const Vec3 &P, const Vec3 &Q, const Vec3 &V) { Vec3 q = Q, v = V; return doesPointLieOnLine(P, q, v); }
bool isPointNearLine(const Vec3 &P, const Vec3 &Q, const Vec3 &V) { Vec3 q = Q, v = V; return isPointNearLine(P, q, v); }
int sideOfLine3D(const Vec3 &P, const Vec3 &N) { Vec3 p = P, n = N; return sideOfLine3D(p, n); }

// reference symbols with external linkage but no header declaration
extern double triangle_intersection(const Vec3 &orig, const Vec3 &dir, const Vec3 &planeNormal, const Vec3 &v0,
                                    const Vec3 &v1, const Vec3 &v2, double &u, double &v); // object.cpp:29
extern double **FF;                              // radiosity.cpp:10
extern double **K;                               // radiosity.cpp:12
extern double *B;                                // radiosity.cpp:14
extern vector<ObjectNode *> radObjectNodes;      // radiosity.cpp:8

// implemented in ref_driver_raster.cpp (that TU includes the reference's rasterizer.cpp)
extern "C" void ref_raster_details_impl(Scene *scene, Camera *cam, double *out18);
extern "C" int ref_raster_frame(Scene *scene, Camera *cam, int mode, const char *workdir, unsigned char *pixels);

namespace {

inline Vec3 V(const double *p) { return Vec3(p[0], p[1], p[2]); }
inline void put(double *p, const Vec3 &a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }

struct Quiet { // the reference chats on cout/cerr for every file and scan line
	std::streambuf *o, *e;
	Quiet() : o(std::cout.rdbuf(nullptr)), e(std::cerr.rdbuf(nullptr)) {}
	~Quiet() { std::cout.rdbuf(o); std::cerr.rdbuf(e); std::cout.clear(); std::cerr.clear(); }
};

double now_s() {
	return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

} // namespace

extern "C" {

// ---------------------------------------------------------------- Möller–Trumbore
// G1: the reference's own triangle_intersection on n independent (ray, triangle) pairs.
// u, v are preset to 0 because the reference leaves them untouched on early returns.
void ref_mt_batch(long n, const double *orig, const double *dir, const double *v0, const double *v1, const double *v2,
                  double *t, double *u, double *v) {
	Vec3 dummyNormal;
	for (long i = 0; i < n; ++i) {
		double uu = 0.0, vv = 0.0;
		t[i] = triangle_intersection(V(orig + 3 * i), V(dir + 3 * i), dummyNormal, V(v0 + 3 * i), V(v1 + 3 * i),
		                             V(v2 + 3 * i), uu, vv);
		u[i] = uu;
		v[i] = vv;
	}
}

// ---------------------------------------------------------------- scene
// workdir must contain scenes/ (base.h:48) and be writable (BuildBSP drops *.dot there).
void *ref_scene_load(const char *workdir, const char *sdf_stem) {
	Quiet q;
	if (chdir(workdir) != 0) return nullptr;
	RefScene *s = new RefScene();
	s->scene.rasterize = new Rasterizer();
	string fname = DefaultScene + sdf_stem; // main.cpp:13-21
	string fobj = DefaultScene + sdf_stem;
	if (!s->scene.BuildScene(fname, fobj, s->cam)) { delete s; return nullptr; }
	return s;
}

// Scene straight from flat arrays (the GPU box has no scene files): one Object per
// object id, triangles in input order; face normal as the reference computes it for
// faces without vertex normals, params.cpp:199-208.  light may be NULL.
void *ref_scene_from_triangles(const double *xyz, const uint32_t *obj, long n, int n_obj, const double *light) {
	Quiet q;
	RefScene *s = new RefScene();
	s->scene.rasterize = new Rasterizer();
	Material *mat = new Material();
	mat->diffuse = White; mat->ambient = Black; mat->emission = Black;
	for (int k = 0; k < n_obj; ++k) {
		Object *o = new Object();
		o->nameOfObject = "obj" + std::to_string(k);
		o->material = mat;
		s->scene.sceneObjects.push_back(o);
	}
	for (long i = 0; i < n; ++i) {
		Object *o = s->scene.sceneObjects[obj ? obj[i] : 0];
		o->triangles.push_back(BSP_Node(BSP_Node::numOfTriangle++));
		BSP_Node &t = o->triangles.back();
		t.triVert1 = V(xyz + 9 * i); t.triVert2 = V(xyz + 9 * i + 3); t.triVert3 = V(xyz + 9 * i + 6);
		t.triNormal = Unit((t.triVert2 - t.triVert1) ^ (t.triVert3 - t.triVert2));
		o->points.push_back(t.triVert1); o->points.push_back(t.triVert2); o->points.push_back(t.triVert3);
	}
	if (light) {
		SphereLight *L = new SphereLight(V(light), 1.0);
		Material *lm = new Material(*mat);
		lm->emission = White;
		L->material = lm;
		s->scene.lights.push_back(L);
	}
	return s;
}

// 0 = built, 1 = the reference threw (scene.cpp:640,678,684 throw const char*), 2 = other exception
int ref_build_bsp(void *h, const char *workdir) {
	Quiet q;
	RefScene *s = (RefScene *)h;
	if (workdir && chdir(workdir) != 0) return 3;
	try {
		s->scene.BuildBSP();
	} catch (const char *) {
		return 1;
	} catch (...) {
		return 2;
	}
	s->bsp_built = true;
	return 0;
}

long ref_num_triangles(void *h) {
	long n = 0;
	for (Object *o : ((RefScene *)h)->scene.sceneObjects) n += (long)o->triangles.size();
	return n;
}
int ref_num_objects(void *h) { return (int)((RefScene *)h)->scene.sceneObjects.size(); }
int ref_num_lights(void *h) { return (int)((RefScene *)h)->scene.NumLights(); }

// original triangles flattened in object order = the "triangle id" contract
void ref_get_triangles(void *h, double *xyz, uint32_t *obj, double *normal) {
	long i = 0;
	uint32_t k = 0;
	for (Object *o : ((RefScene *)h)->scene.sceneObjects) {
		for (BSP_Node &t : o->triangles) {
			put(xyz + 9 * i, t.triVert1); put(xyz + 9 * i + 3, t.triVert2); put(xyz + 9 * i + 6, t.triVert3);
			obj[i] = k;
			put(normal + 3 * i, t.triNormal);
			++i;
		}
		++k;
	}
}

// eye[3] lookat[3] up[3] vpdist xwin[2] ywin[2]; res = {x_res, y_res}
void ref_get_camera(void *h, double *c, uint32_t *res) {
	Camera &cam = ((RefScene *)h)->cam;
	put(c, cam.eye); put(c + 3, cam.lookat); put(c + 6, cam.up);
	c[9] = cam.vpdist; c[10] = cam.x_win.min; c[11] = cam.x_win.max; c[12] = cam.y_win.min; c[13] = cam.y_win.max;
	res[0] = cam.x_res; res[1] = cam.y_res;
}

// centre of the first light's slab box, as every caller computes it (shader.cpp:120-124, radiosity.cpp:270-272)
int ref_light_center(void *h, double *c) {
	RefScene *s = (RefScene *)h;
	if (s->scene.NumLights() == 0) return 1;
	put(c, Center(GetBox(*s->scene.GetLight(0))));
	return 0;
}

// G,U,R,O,dR,dU (18 doubles) from the reference's own RasterDetails constructor, rasterizer.cpp:40-68
void ref_raster_details(void *h, double *out18) {
	RefScene *s = (RefScene *)h;
	ref_raster_details_impl(&s->scene, &s->cam, out18);
}

// the reference's primary-ray direction, rasterizer.cpp:230, through its own Vec3 operators.
// pix = n (i=row, j=col) pairs.
void ref_primary_dirs(const double *rd18, long n, const uint32_t *pix, double *dir) {
	Vec3 O = V(rd18 + 9), dR = V(rd18 + 12), dU = V(rd18 + 15);
	for (long k = 0; k < n; ++k) {
		unsigned i = pix[2 * k], j = pix[2 * k + 1];
		put(dir + 3 * k, Unit(O + (j + 0.5) * dR - (i + 0.5) * dU));
	}
}

// Scene::Cast (needs ref_build_bsp).  obj = index into sceneObjects or -1.
// returns seconds spent inside the Cast loop (CPU baseline timing).
double ref_cast(void *h, long n, const double *o, const double *d, uint8_t *hit, double *dist, double *point,
                double *normal, int32_t *obj, float *bu, float *bv) {
	RefScene *s = (RefScene *)h;
	Ray ray;
	ray.type = generic_ray;
	ray.generation = 1;
	double t0 = now_s();
	for (long k = 0; k < n; ++k) {
		ray.origin = V(o + 3 * k);
		ray.direction = V(d + 3 * k);
		HitInfo hi;
		hi.ignore = NULL;
		hi.distance = Infinity; // scene.cpp:131-132
		bool f = s->scene.Cast(ray, hi);
		hit[k] = f ? 1 : 0;
		if (f) {
			dist[k] = hi.distance;
			put(point + 3 * k, hi.point);
			put(normal + 3 * k, hi.normal);
			int idx = -1, c = 0;
			for (Object *ob : s->scene.sceneObjects) { if (ob == hi.object) idx = c; ++c; }
			obj[k] = idx;
			bu[k] = hi.BayCentricU;
			bv[k] = hi.BayCentricV;
		} else {
			dist[k] = 0.0; obj[k] = -1; bu[k] = bv[k] = 0.f;
			point[3 * k] = point[3 * k + 1] = point[3 * k + 2] = 0.0;
			normal[3 * k] = normal[3 * k + 1] = normal[3 * k + 2] = 0.0;
		}
	}
	return now_s() - t0;
}

// Shader::findOcclusionNew (shader.cpp:140-232) at n surface points: out[k] = 1 − occlusions/nrays.  The reference
// seeds its mt19937 from std::random_device inside the call, so the value is a Monte-Carlo estimate; callers
// average many rays.  Needs ref_build_bsp (it calls Scene::Cast).
void ref_ao_occlusion(void *h, long n, const double *P, const double *N, double radius, int nrays, double *out) {
	RefScene *s = (RefScene *)h;
	Shader sh;
	for (long k = 0; k < n; ++k) {
		HitInfo hit;
		hit.point = V(P + 3 * k);
		hit.normal = V(N + 3 * k);
		Color c;
		out[k] = sh.findOcclusionNew(s->scene, hit, c, radius, nrays);
	}
}

// Shader::generateSoftShadows (shader.cpp:41-103) at n surface points: hit.point = P[k], hit.object = sceneObjects[obj[k]];
// colour White in -> out[k] = the factor 1 − contributors/100 (:100).  Needs a light and a built scene.
void ref_soft_shadow(void *h, long n, const double *P, const int32_t *obj, double *out) {
	RefScene *s = (RefScene *)h;
	Shader sh;
	for (long k = 0; k < n; ++k) {
		HitInfo hit;
		hit.point = V(P + 3 * k);
		hit.object = s->scene.sceneObjects[obj[k]];
		Color c = White;
		out[k] = sh.generateSoftShadows(s->scene, hit, c).red;
	}
}

// Brute force with the reference's triangle_intersection over the ORIGINAL triangles in
// flattened order, keeping the strict minimum of the positive t (first wins on ties) —
// the semantics of intersectTreeTraversalNoAccel (object.cpp:143-168) + Scene::Cast's
// strict '<' over objects (scene.cpp:197).  tri = -1 on a miss.  returns seconds.
double ref_brute(void *h, long n, const double *o, const double *d, int32_t *tri, double *t, double *u, double *v) {
	RefScene *s = (RefScene *)h;
	std::vector<const BSP_Node *> flat;
	for (Object *ob : s->scene.sceneObjects)
		for (BSP_Node &b : ob->triangles) flat.push_back(&b);
	double t0 = now_s();
	for (long k = 0; k < n; ++k) {
		Vec3 O = V(o + 3 * k), D = V(d + 3 * k);
		double best = Infinity, bu = 0, bv = 0;
		int32_t bi = -1;
		for (size_t i = 0; i < flat.size(); ++i) {
			double uu = 0, vv = 0;
			double tt = triangle_intersection(O, D, flat[i]->triNormal, flat[i]->triVert1, flat[i]->triVert2,
			                                  flat[i]->triVert3, uu, vv);
			if (tt > 0.0 && best > tt) { best = tt; bi = (int32_t)i; bu = uu; bv = vv; }
		}
		tri[k] = bi; t[k] = bi >= 0 ? best : 0.0; u[k] = bu; v[k] = bv;
	}
	return now_s() - t0;
}

// shadow ray of generateNormalShadows (shader.cpp:116-127): origin P + 0.01·n, dir Unit(C − P)
void ref_shadow_rays(long n, const double *P, const double *N, const double *C, double *o, double *d) {
	Vec3 c = V(C);
	for (long k = 0; k < n; ++k) {
		Vec3 p = V(P + 3 * k), nn = V(N + 3 * k);
		put(d + 3 * k, Unit(c - p));
		put(o + 3 * k, p + (nn * 0.01));
	}
}

// ---------------------------------------------------------------- shading (Scene::Trace / Shader::Shade)
// A scene of n one-triangle Objects: Object::root is the triangle itself (a BSP leaf), so Object::Intersect
// (object.cpp:170-184) runs the reference's Möller–Trumbore on the ORIGINAL triangle and Scene::Cast's strict '<'
// over sceneObjects (scene.cpp:193-205) is brute force — no BSP build, no fragments.  mats: n_mat × 18 doubles
// (diffuse, specular, ambient, reflectivity, translucency, Phong_exp, ref_index, roughness); lights: n_l × 6
// (centre of a radius-1 SphereLight, emission).  normals may be NULL (face normal as params.cpp:199-208).
void *ref_scene_one_tri_objects(const double *xyz, const double *normals, long n, const uint32_t *mat_of, const double *mats,
                                int n_mat, const double *lights, int n_l) {
	Quiet q;
	RefScene *s = new RefScene();
	s->scene.rasterize = new Rasterizer();
	std::vector<Material *> M;
	for (int k = 0; k < n_mat; ++k) {
		const double *m = mats + 18 * k;
		Material *mat = new Material();
		mat->diffuse = Color(m[0], m[1], m[2]); mat->specular = Color(m[3], m[4], m[5]);
		mat->ambient = Color(m[6], m[7], m[8]); mat->reflectivity = Color(m[9], m[10], m[11]);
		mat->translucency = Color(m[12], m[13], m[14]); mat->emission = Black;
		mat->Phong_exp = m[15]; mat->ref_index = m[16]; mat->microFacetsRoughness = m[17];
		M.push_back(mat);
	}
	for (long i = 0; i < n; ++i) {
		Object *o = new Object();
		o->nameOfObject = "tri" + std::to_string(i);
		o->material = M[mat_of[i]];
		o->triangles.reserve(1);
		o->triangles.push_back(BSP_Node(BSP_Node::numOfTriangle++));
		BSP_Node &t = o->triangles.back();
		t.triVert1 = V(xyz + 9 * i); t.triVert2 = V(xyz + 9 * i + 3); t.triVert3 = V(xyz + 9 * i + 6);
		t.triNormal = normals ? V(normals + 3 * i) : Unit((t.triVert2 - t.triVert1) ^ (t.triVert3 - t.triVert2));
		t.left = t.right = NULL;
		o->root = &t;
		s->scene.sceneObjects.push_back(o);
	}
	for (int k = 0; k < n_l; ++k) {
		SphereLight *L = new SphereLight(V(lights + 6 * k), 1.0);
		Material *lm = new Material();
		lm->emission = Color(lights[6 * k + 3], lights[6 * k + 4], lights[6 * k + 5]);
		L->material = lm;
		s->scene.lights.push_back(L);
	}
	s->bsp_built = true;
	return s;
}

// per sceneObject: the 18 material doubles of ref_scene_one_tri_objects; per light: Center(GetBox) and emission
void ref_get_materials(void *h, double *out) {
	long k = 0;
	for (Object *o : ((RefScene *)h)->scene.sceneObjects) {
		const Material &m = *o->material;
		double *p = out + 18 * k++;
		const Color *c[5] = {&m.diffuse, &m.specular, &m.ambient, &m.reflectivity, &m.translucency};
		for (int a = 0; a < 5; ++a) { p[3 * a] = c[a]->red; p[3 * a + 1] = c[a]->green; p[3 * a + 2] = c[a]->blue; }
		p[15] = m.Phong_exp; p[16] = m.ref_index; p[17] = m.microFacetsRoughness;
	}
}
void ref_get_lights(void *h, double *out) {
	RefScene *s = (RefScene *)h;
	for (unsigned i = 0; i < s->scene.NumLights(); ++i) {
		const PrimitiveObject *L = s->scene.GetLight(i);
		put(out + 6 * i, Center(GetBox(*L)));
		out[6 * i + 3] = L->material->emission.red; out[6 * i + 4] = L->material->emission.green;
		out[6 * i + 5] = L->material->emission.blue;
	}
}
// overwrite every sceneObject's material (18 doubles each, as ref_scene_one_tri_objects)
void ref_set_materials(void *h, const double *mats) {
	long k = 0;
	for (Object *o : ((RefScene *)h)->scene.sceneObjects) {
		const double *m = mats + 18 * k++;
		Material *mat = new Material();
		mat->diffuse = Color(m[0], m[1], m[2]); mat->specular = Color(m[3], m[4], m[5]);
		mat->ambient = Color(m[6], m[7], m[8]); mat->reflectivity = Color(m[9], m[10], m[11]);
		mat->translucency = Color(m[12], m[13], m[14]); mat->emission = Black;
		mat->Phong_exp = m[15]; mat->ref_index = m[16]; mat->microFacetsRoughness = m[17];
		o->material = mat;
	}
}
// camera of a scene built from arrays; then the reference's own raster loops (ref_driver_raster.cpp)
void ref_set_camera14(void *h, const double *c14, uint32_t x_res, uint32_t y_res) {
	Camera &cam = ((RefScene *)h)->cam;
	cam.eye = V(c14); cam.lookat = V(c14 + 3); cam.up = V(c14 + 6);
	cam.vpdist = c14[9]; cam.x_win.min = c14[10]; cam.x_win.max = c14[11]; cam.y_win.min = c14[12]; cam.y_win.max = c14[13];
	cam.x_res = x_res; cam.y_res = y_res;
}
int ref_raster(void *h, int mode, const char *workdir, unsigned char *pixels) {
	RefScene *s = (RefScene *)h;
	return ref_raster_frame(&s->scene, &s->cam, mode, workdir, pixels);
}
void ref_set_max_depth(void *h, unsigned d) { ((RefScene *)h)->scene.max_tree_depth = d; }

// Scene::Trace (scene.cpp:127-155) on n rays of the given generation (1 = from the eye, rasterizer.cpp:217);
// rgb: n × 3 doubles, before ToneMap.  Ambient occlusion inside Shade draws from an unseeded generator
// (shader.cpp:148-150): results are reproducible only where the material's ambient is 0.  returns seconds.
double ref_trace(void *h, long n, const double *o, const double *d, unsigned generation, double *rgb) {
	RefScene *s = (RefScene *)h;
	Ray ray;
	ray.type = generic_ray;
	ray.generation = generation;
	double t0 = now_s();
	for (long k = 0; k < n; ++k) {
		ray.origin = V(o + 3 * k);
		ray.direction = V(d + 3 * k);
		Color c = s->scene.Trace(ray);
		rgb[3 * k] = c.red; rgb[3 * k + 1] = c.green; rgb[3 * k + 2] = c.blue;
	}
	return now_s() - t0;
}

// ---------------------------------------------------------------- radiosity
// Sets the public fields the constructor sets (radiosity.cpp:23-44) and runs
// findIntersections + buildInitialQuadTree.  Returns numOfElements (= BSP leaves).
int ref_rad_setup(void *h) {
	Quiet q;
	RefScene *s = (RefScene *)h;
	if (!s->bsp_built) return -1;
	Radiosity *R = new Radiosity();
	R->scene = &s->scene;
	R->cam = &s->cam;
	R->numOfElements = 0;
	R->quadTreeRoot = NULL;
	R->radiosityHelper = new Radiosity_Helper(&s->scene);
	R->tempQuadVector = new vector<QuadTreeNode *>(0);
	for (auto v : radObjectNodes) delete v;
	radObjectNodes.clear();
	for (Object *ob : s->scene.sceneObjects) {
		ObjectNode *on = new ObjectNode();
		radObjectNodes.push_back(on);
		on->object = ob;
		on->aabbInfo = GetBoxPolygon(ob);
	}
	R->findIntersections();
	R->buildInitialQuadTree();
	s->rad = R;
	s->elems.clear();
	for (auto &v : R->quadTreeRoot->children) R->returnFilledElementsOfObject(v, s->elems);
	return R->numOfElements;
}

// element i ↔ B[i], FF[i][·] (radiosity.cpp:373-385): vertices, triNormal, object index
void ref_rad_get_elements(void *h, double *verts, double *normal, int32_t *obj) {
	RefScene *s = (RefScene *)h;
	for (size_t i = 0; i < s->elems.size(); ++i) {
		QuadTreeNode *e = s->elems[i];
		put(verts + 9 * i, e->triVert1); put(verts + 9 * i + 3, e->triVert2); put(verts + 9 * i + 6, e->triVert3);
		put(normal + 3 * i, e->triNormal);
		int idx = -1, c = 0;
		for (Object *ob : s->scene.sceneObjects) { if (ob == e->object) idx = c; ++c; }
		obj[i] = idx;
	}
}

// isVisibleXiToXj (radiosity.cpp:231-262) on injected points
void ref_rad_visible(void *h, long m, const int32_t *ei, const int32_t *ej, const double *xi, const double *xj,
                     uint8_t *vis) {
	RefScene *s = (RefScene *)h;
	for (long k = 0; k < m; ++k)
		vis[k] = s->rad->isVisibleXiToXj(s->elems[ei[k]], s->elems[ej[k]], V(xi + 3 * k), V(xj + 3 * k)) ? 1 : 0;
}

// the reference's own Monte-Carlo terms (unseeded RNG → statistical comparison only)
double ref_rad_ff_term(void *h, int i, int j) {
	RefScene *s = (RefScene *)h;
	return s->rad->findFormFactorTerm(s->elems[i], s->elems[j]);
}
double ref_rad_ff_light_term(void *h, int i) {
	RefScene *s = (RefScene *)h;
	return s->rad->findFormFactorTermWithLight(s->elems[i]);
}

// castFromElementToElement as findFormFactorTermWithLight drives it (radiosity.cpp:279-290):
// origin xi + 1e-5·n_i, direction Unit(C − xi); returns 1 if anything is hit.
void ref_rad_light_blocked(void *h, long m, const int32_t *ei, const double *xi, uint8_t *blocked) {
	RefScene *s = (RefScene *)h;
	Vec3 centerEx = Center(GetBox(*s->scene.GetLight(0)));
	for (long k = 0; k < m; ++k) {
		QuadTreeNode *e = s->elems[ei[k]];
		Vec3 x = V(xi + 3 * k);
		Ray ray;
		ray.direction = Unit(centerEx - x);
		ray.origin = x + (Epsilon * e->triNormal);
		ray.type = generic_ray;
		ray.generation = 1;
		HitInfo hi;
		hi.ignore = NULL;
		hi.distance = Infinity;
		blocked[k] = s->rad->castFromElementToElement(ray, hi, e, NULL, x, centerEx) ? 1 : 0;
	}
}

// findFormFactor (radiosity.cpp:335-467): allocates FF, K and fills them with the
// reference's own random estimate.  returns seconds.
double ref_rad_find_form_factor(void *h) {
	Quiet q;
	RefScene *s = (RefScene *)h;
	double t0 = now_s();
	s->rad->findFormFactor();
	return now_s() - t0;
}

int ref_rad_n(void *h) {
	RefScene *s = (RefScene *)h;
	return s->rad->numOfElements + (int)s->scene.NumLights();
}

void ref_rad_get_FF(void *h, double *out) {
	int n = ref_rad_n(h);
	for (int i = 0; i < n; ++i)
		for (int j = 0; j < n; ++j) out[(size_t)i * n + j] = FF[i][j];
}

// Inject a form-factor matrix (n×n incl. the light column/row) into the reference's
// globals and rebuild K exactly as findFormFactor does after filling FF
// (radiosity.cpp:413-428: element rows 1−0.7·FF, light row δ − FF).
void ref_rad_set_FF(void *h, const double *F) {
	RefScene *s = (RefScene *)h;
	int ne = s->rad->numOfElements, n = ref_rad_n(h);
	for (int i = 0; i < n; ++i)
		for (int j = 0; j < n; ++j) FF[i][j] = F[(size_t)i * n + j];
	for (int i = 0; i < ne; ++i)
		for (int j = 0; j < ne; ++j) K[i][j] = (i == j ? 1.0 : 0.0) - (REFLECTIVITY_INDEX * FF[i][j]);
	for (int i = 0; i < ne; ++i) K[i][n - 1] = (i == n - 1 ? 1.0 : 0.0) - (REFLECTIVITY_INDEX * FF[i][n - 1]);
	for (int i = 0; i < n; ++i) K[n - 1][i] = (n - 1 == i ? 1.0 : 0.0) - FF[n - 1][i];
}

// progressiveRefinement (radiosity.cpp:483-563) on whatever FF/K hold; returns seconds
double ref_rad_solve(void *h) {
	RefScene *s = (RefScene *)h;
	double t0 = now_s();
	s->rad->progressiveRefinement();
	return now_s() - t0;
}

void ref_rad_get_B(void *h, double *out) {
	int n = ref_rad_n(h);
	for (int i = 0; i < n; ++i) out[i] = B[i];
}

// setRadiosityForAllElements (radiosity.cpp:122-134) then read QuadTreeNode::radiosityValue back
void ref_rad_scatter_B(void *h, double *out) {
	RefScene *s = (RefScene *)h;
	s->rad->setRadiosityForAllElements(s->rad->quadTreeRoot);
	for (size_t i = 0; i < s->elems.size(); ++i) out[i] = s->elems[i]->radiosityValue;
}

// Makes Shader::Shade take its radiosity branch (shader.cpp:431-478): scene.radiosity != NULL.
//   mode 0: detach;  1: the Radiosity of ref_rad_setup, tempQuadVector = its element list (what main.cpp's radiosity run has);
//   2: a bare Radiosity (helper + empty element list) for scenes without a BSP: trace_ray then answers Green, and the
//      branch's colour does not depend on it (the uses of radiosityAmbient and occlusion are commented out, :457-465).
int ref_rad_attach(void *h, int mode) {
	RefScene *s = (RefScene *)h;
	if (mode == 0) { s->scene.radiosity = NULL; return 0; }
	if (mode == 1) {
		if (!s->rad) return -1;
		*s->rad->tempQuadVector = s->elems;
		s->scene.radiosity = s->rad;
		return 0;
	}
	Radiosity *R = new Radiosity();
	R->scene = &s->scene;
	R->cam = &s->cam;
	R->numOfElements = 0;
	R->quadTreeRoot = NULL;
	R->radiosityHelper = new Radiosity_Helper(&s->scene);
	R->tempQuadVector = new vector<QuadTreeNode *>(0);
	s->scene.radiosity = R;
	return 0;
}

// ---- radiosity display path (Rasterizer::Radiosity_Raster, rasterizer.cpp:408-470)
// QuadTreeNode::radiosityValue of every element (what setRadiosityForAllElements would write)
void ref_rad_set_element_radiosity(void *h, const double *B) {
	RefScene *s = (RefScene *)h;
	for (size_t i = 0; i < s->elems.size(); ++i) s->elems[i]->radiosityValue = B[i];
}
// Radiosity_Helper::trace_ray (radiosity_helper.cpp:318-365) over the element list, as Radiosity_Raster calls it
void ref_rad_trace_ray(void *h, long n, const double *o, const double *d, double *rgb) {
	RefScene *s = (RefScene *)h;
	Ray ray;
	ray.type = generic_ray;
	ray.generation = 1;
	for (long k = 0; k < n; ++k) {
		ray.origin = V(o + 3 * k);
		ray.direction = V(d + 3 * k);
		Color c = s->rad->radiosityHelper->trace_ray(ray, s->elems);
		rgb[3 * k] = c.red; rgb[3 * k + 1] = c.green; rgb[3 * k + 2] = c.blue;
	}
}
// Fills QuadTreeNode::nextAdj (the reference's own build of it is commented out, radiosity.cpp:167-199) with m records
// "element ei[k] shares the point pts[3k..] with element ej[k]"; m = 0 clears every list.
void ref_rad_set_adjacency(void *h, long m, const int *ei, const int *ej, const double *pts) {
	RefScene *s = (RefScene *)h;
	if (m == 0) for (QuadTreeNode *e : s->elems) e->nextAdj.clear();
	for (long k = 0; k < m; ++k) s->elems[ei[k]]->nextAdj[s->elems[ej[k]]].insert(V(pts + 3 * k));
}
// returnQuadTreeColor (radiosity_helper.h:156-268) evaluated AT the three vertices of every element: there the
// barycentric weights are exactly (1,0,0), (0,1,0), (0,0,1), so the result is the interpolated radiosity the reference
// attaches to that vertex (shared-vertex averages over nextAdj) — out: n_elements × 3
void ref_rad_vertex_radiosity(void *h, double *out) {
	RefScene *s = (RefScene *)h;
	for (size_t i = 0; i < s->elems.size(); ++i) {
		QuadTreeNode *e = s->elems[i];
		Vec3 v[3] = {e->triVert1, e->triVert2, e->triVert3};
		for (int k = 0; k < 3; ++k) out[3 * i + k] = returnQuadTreeColor(v[k], e).red;
	}
}

// residual the reference's stop test looks at (radiosity.cpp:469-481): max_i |Σ_j K_ij B_j − E_i|
double ref_rad_residual(void *h) {
	int n = ref_rad_n(h);
	double worst = 0;
	for (int i = 0; i < n; ++i) {
		double sum = 0.0;
		for (int j = 0; j < n; ++j) sum += K[i][j] * B[j];
		double e = (i == n - 1) ? (SOLAR_RADIOSITY_POWER) : 0.0;
		double r = std::abs(sum - e);
		if (r > worst) worst = r;
	}
	return worst;
}

} // extern "C"
