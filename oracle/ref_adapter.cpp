// TEST INFRASTRUCTURE — not part of the product.
//
// The drop-in proof: include/rtmi_adapter.hpp (the product's C++ host glue) instantiated with the COMPILED
// REFERENCE's own types (Render2/ray_tracer.h: Scene, Object, BSP_Node, Ray, HitInfo, QuadTreeNode) and run
// beside the reference's own Scene::Cast / Radiosity on the same objects.  Linked, with the reference's objects
// and ref_driver.cpp, into oracle/_ref/libref_adapter.so (needs librtmi.so and the HIP runtime, i.e. a GPU box);
// only tests/ load it.
#include "ray_tracer.h"
#include "util.h"
#include "radiosity_helper.h"
#include "ppm_image.h"
#include "ref_scene.h"

#include "../include/rtmi_adapter.hpp"

#include <cstdint>
#include <cstring>

namespace {
typedef rtmi_host::SceneBinding<Scene, Object, BSP_Node> Binding;
typedef rtmi_host::RadiosityBinding<QuadTreeNode> RadBinding;
typedef rtmi_host::ShaderBinding<Scene, Object, BSP_Node> ShadeBinding;
inline Vec3 V(const double *p) { return Vec3(p[0], p[1], p[2]); }
inline void put(double *p, const Vec3 &a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
} // namespace

extern "C" {

const char *ref_adapter_error() { return rtmi_last_error(); }

// Scene::BuildBSP through the adapter: uploads the reference scene's original triangles
void *ref_adapter_bind(void *h) {
	RefScene *s = (RefScene *)h;
	Binding *b = new Binding();
	if (!b->BuildBSP(s->scene)) { delete b; return nullptr; }
	return b;
}
void ref_adapter_unbind(void *b) { delete (Binding *)b; }

// the same output layout as ref_cast (ref_driver.cpp), produced by the adapter's batch Cast into the reference's
// HitInfo structs; additionally tri = flattened index of HitInfo::tri among the original triangles.
int ref_adapter_cast(void *h, void *bind, long n, const double *o, const double *d, uint8_t *hit, double *dist, double *point,
                     double *normal, int32_t *obj, float *bu, float *bv, int32_t *tri) {
	RefScene *s = (RefScene *)h;
	Binding *b = (Binding *)bind;
	std::vector<Ray> rays(n);
	std::vector<HitInfo> hits(n);
	for (long k = 0; k < n; ++k) {
		rays[k].origin = V(o + 3 * k);
		rays[k].direction = V(d + 3 * k);
		rays[k].type = generic_ray;
		rays[k].generation = 1;
		hits[k].ignore = NULL;
		hits[k].distance = Infinity;
	}
	try {
		b->Cast(rays.data(), (size_t)n, hits.data(), hit);
	} catch (const std::exception &) {
		return -1;
	}
	std::vector<const BSP_Node *> flat;
	for (Object *ob : s->scene.sceneObjects)
		for (BSP_Node &t : ob->triangles) flat.push_back(&t);
	for (long k = 0; k < n; ++k) {
		if (hit[k]) {
			const HitInfo &hi = hits[k];
			dist[k] = hi.distance;
			put(point + 3 * k, hi.point);
			put(normal + 3 * k, hi.normal);
			int idx = -1, c = 0;
			for (Object *ob : s->scene.sceneObjects) { if (ob == hi.object) idx = c; ++c; }
			obj[k] = idx;
			bu[k] = hi.BayCentricU;
			bv[k] = hi.BayCentricV;
			tri[k] = -1;
			for (size_t i = 0; i < flat.size(); ++i) if (flat[i] == hi.tri) { tri[k] = (int32_t)i; break; }
		} else {
			dist[k] = 0.0; obj[k] = -1; bu[k] = bv[k] = 0.f; tri[k] = -1;
			point[3 * k] = point[3 * k + 1] = point[3 * k + 2] = 0.0;
			normal[3 * k] = normal[3 * k + 1] = normal[3 * k + 2] = 0.0;
		}
	}
	return 0;
}

// per-ray signature: bool Scene::Cast(const Ray&, HitInfo&)
int ref_adapter_cast_one(void *bind, const double *o, const double *d, double *dist) {
	Binding *b = (Binding *)bind;
	Ray r;
	r.origin = V(o); r.direction = V(d);
	HitInfo hi;
	hi.ignore = NULL; hi.distance = Infinity;
	const bool f = b->Cast(r, hi);
	*dist = f ? hi.distance : 0.0;
	return f ? 1 : 0;
}

int ref_adapter_occluded(void *bind, long n, const double *o, const double *d, const double *tmax, uint8_t *occ) {
	Binding *b = (Binding *)bind;
	std::vector<Ray> rays(n);
	for (long k = 0; k < n; ++k) { rays[k].origin = V(o + 3 * k); rays[k].direction = V(d + 3 * k); }
	try {
		b->Occluded(rays.data(), tmax, (size_t)n, occ);
	} catch (const std::exception &) {
		return -1;
	}
	return 0;
}

// Rasterizer::Rasterize's work through the adapter: the reference's Scene (materials, lights, max_tree_depth) and
// Camera -> a PPM_Image filled on the GPU; pixels = x_res·y_res·3 bytes in the reference's row order.  Also Scene::Trace
// on the frame's first n_probe primary rays -> probe_rgb (n_probe × 3), through the batch Trace().
int ref_adapter_rasterize(void *h, void *bind, uint32_t ao_rays, uint64_t seed, uint8_t *pixels, long n_probe, const double *probe_o,
                          const double *probe_d, double *probe_rgb) {
	RefScene *s = (RefScene *)h;
	try {
		ShadeBinding sb;
		if (!sb.bind(*(Binding *)bind, s->scene, ao_rays, 2.0, seed)) return -1;
		PPM_Image I(s->cam.x_res, s->cam.y_res);
		sb.Rasterize(s->cam, I);
		for (unsigned i = 0; i < s->cam.y_res; ++i)
			for (unsigned j = 0; j < s->cam.x_res; ++j) {
				const Pixel &p = I(i, j);
				uint8_t *q = pixels + 3 * ((size_t)i * s->cam.x_res + j);
				q[0] = p.r; q[1] = p.g; q[2] = p.b;
			}
		std::vector<Ray> rays(n_probe);
		std::vector<Color> cols(n_probe);
		for (long k = 0; k < n_probe; ++k) {
			rays[k].origin = V(probe_o + 3 * k); rays[k].direction = V(probe_d + 3 * k);
			rays[k].type = generic_ray; rays[k].generation = 1;
		}
		sb.Trace(rays.data(), (size_t)n_probe, cols.data());
		for (long k = 0; k < n_probe; ++k) { probe_rgb[3 * k] = cols[k].red; probe_rgb[3 * k + 1] = cols[k].green; probe_rgb[3 * k + 2] = cols[k].blue; }
		return 0;
	} catch (const std::exception &) {
		return -1;
	}
}

// Radiosity through the adapter on the reference's own element list (needs ref_rad_setup): fills FF_out (n × n doubles,
// what the reference's global FF would hold), B_out (n) and writes QuadTreeNode::radiosityValue; returns n or -1.
int ref_adapter_radiosity(void *h, const double *light_center, uint32_t samples, uint32_t light_samples, uint64_t seed,
                          double *FF_out, double *B_out, double *elem_radiosity, uint32_t *sweeps) {
	RefScene *s = (RefScene *)h;
	try {
		RadBinding rb;
		if (!rb.setElements(s->elems, light_center)) return -1;
		rb.findFormFactor(samples, light_samples, seed, 0);
		if (FF_out) rb.downloadFF(FF_out);
		rtmi_solve_stats st;
		std::vector<double> B = rb.progressiveRefinement(REFLECTIVITY_INDEX, SOLAR_RADIOSITY_POWER, 1e-7f, &st);
		rb.setRadiosityForAllElements();
		std::memcpy(B_out, B.data(), sizeof(double) * B.size());
		for (size_t i = 0; i < s->elems.size(); ++i) elem_radiosity[i] = s->elems[i]->radiosityValue;
		if (sweeps) *sweeps = st.sweeps;
		return (int)rb.n();
	} catch (const std::exception &) {
		return -1;
	}
}

// Rasterizer::Radiosity_Raster's pixel loop through the adapter, on the reference's element list (needs ref_rad_setup)
// with whatever QuadTreeNode::radiosityValue holds: pixels = x_res·y_res·3 ToneMapRadiosity bytes.
int ref_adapter_radiosity_raster(void *h, uint8_t *pixels) {
	RefScene *s = (RefScene *)h;
	try {
		RadBinding rb;
		if (!rb.setElements(s->elems, nullptr)) return -1;
		PPM_Image I(s->cam.x_res, s->cam.y_res);
		rb.Radiosity_Raster(s->cam, I);
		for (unsigned i = 0; i < s->cam.y_res; ++i)
			for (unsigned j = 0; j < s->cam.x_res; ++j) {
				const Pixel &p = I(i, j);
				uint8_t *q = pixels + 3 * ((size_t)i * s->cam.x_res + j);
				q[0] = p.r; q[1] = p.g; q[2] = p.b;
			}
		return 0;
	} catch (const std::exception &) {
		return -1;
	}
}

} // extern "C"
