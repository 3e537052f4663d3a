// rtmi_adapter.hpp — C++ host glue between the RayTracer host driver and the C ABI of librtmi.so.
//
// Header-only and templated on the reference's own types, so it includes nothing of the reference: the host
// driver instantiates it with its Scene / Object / BSP_Node / Ray / HitInfo / QuadTreeNode (Render2/ray_tracer.h)
// after including its own headers.  Only member NAMES of those types are used (cited below).  Compile the
// translation unit that uses it with -ffp-contract=off: hit points are evaluated in the reference's operation
// order and must round the same way.
//
//   rtmi_host::SceneBinding   Scene::BuildBSP (scene.cpp:463)  -> rtmi_scene_create over the ORIGINAL triangles
//                             Scene::Cast (scene.cpp:159-207) + updateHitInfo (object.cpp:13-24) -> batch Cast()
//                             Cast as a predicate (shader.cpp:131,226)                         -> Occluded()
//   rtmi_host::ShaderBinding  Scene::Trace + Shader::Shade (scene.cpp:127-155, shader.cpp:395-645) and the pixel loop of
//                             Rasterizer::Serial_Normal_Raster (rasterizer.cpp:191-241)            -> Trace(), Rasterize()
//   rtmi_host::RadiosityBinding  findFormFactor / progressiveRefinement / setRadiosityForAllElements
//                             (radiosity.cpp:335-467, 483-563, 122-134) on the element list; Radiosity_Raster
//                             (rasterizer.cpp:408-470) into the reference's PPM_Image
#ifndef RTMI_ADAPTER_HPP
#define RTMI_ADAPTER_HPP
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "rtmi.h"

namespace rtmi_host {

struct Error : std::runtime_error {
	explicit Error(const std::string &what) : std::runtime_error(what + ": " + rtmi_last_error()) {}
};

template <class V3> inline void put3(double *p, const V3 &a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }

// ------------------------------------------------------------------------------------------------ ray queries
template <class SceneT, class ObjectT, class TriT>
class SceneBinding {
public:
	SceneBinding() = default;
	SceneBinding(const SceneBinding &) = delete;
	SceneBinding &operator=(const SceneBinding &) = delete;
	~SceneBinding() { rtmi_scene_destroy(scene_); }

	// Scene::BuildBSP: the triangle id is the position in sceneObjects[k]->triangles, flattened in object order
	// (Render2/ray_tracer.h:200).  Returns false with the text in rtmi_last_error(), as BuildBSP returns bool.
	bool BuildBSP(SceneT &scene) {
		std::vector<double> xyz;
		std::vector<uint32_t> obj;
		tri_of_id_.clear();
		object_of_tri_.clear();
		uint32_t k = 0;
		for (ObjectT *o : scene.sceneObjects) {
			for (TriT &t : o->triangles) {
				double v[9];
				put3(v, t.triVert1); put3(v + 3, t.triVert2); put3(v + 6, t.triVert3);
				xyz.insert(xyz.end(), v, v + 9);
				obj.push_back(k);
				tri_of_id_.push_back(&t);
				object_of_tri_.push_back(o);
			}
			++k;
		}
		rtmi_scene_destroy(scene_);
		scene_ = rtmi_scene_create(xyz.data(), obj.data(), obj.size());
		return scene_ != nullptr;
	}

	rtmi_scene *handle() const { return scene_; }
	size_t triangles() const { return tri_of_id_.size(); }
	const TriT *triangle(size_t id) const { return tri_of_id_[id]; }

	// Scene::Cast for a batch: found[i] = the reference's return value; hits[i] is filled exactly as
	// updateHitInfo + Scene::Cast leave it (object.cpp:13-24, scene.cpp:197-203) and untouched on a miss.
	template <class RayT, class HitInfoT>
	void Cast(const RayT *rays, size_t n, HitInfoT *hits, unsigned char *found) const {
		std::vector<rtmi_ray> r(n);
		std::vector<rtmi_hit> h(n);
		for (size_t i = 0; i < n; ++i) {
			put3(r[i].o, rays[i].origin);
			put3(r[i].d, rays[i].direction);
			r[i].tmax = RTMI_INFINITY;
			r[i].reserved = 0.0;
		}
		if (rtmi_closest_hit(scene_, r.data(), n, h.data()) != 0) throw Error("rtmi_closest_hit");
		for (size_t i = 0; i < n; ++i) {
			found[i] = h[i].tri != RTMI_MISS;
			if (found[i]) fill(rays[i], h[i], hits[i]);
		}
	}

	// the per-ray signature of the reference, bool Scene::Cast(const Ray&, HitInfo&) const (ray_tracer.h:153)
	template <class RayT, class HitInfoT>
	bool Cast(const RayT &ray, HitInfoT &hitinfo) const {
		unsigned char f = 0;
		Cast(&ray, 1, &hitinfo, &f);
		return f != 0;
	}

	// Cast used as a predicate: occluded[i] = Cast(ray) && distance < tmax[i]  (tmax == nullptr: any distance)
	template <class RayT>
	void Occluded(const RayT *rays, const double *tmax, size_t n, unsigned char *occluded) const {
		std::vector<rtmi_ray> r(n);
		for (size_t i = 0; i < n; ++i) {
			put3(r[i].o, rays[i].origin);
			put3(r[i].d, rays[i].direction);
			r[i].tmax = tmax ? tmax[i] : RTMI_INFINITY;
			r[i].reserved = 0.0;
		}
		if (rtmi_any_hit(scene_, r.data(), n, occluded) != 0) throw Error("rtmi_any_hit");
	}

private:
	template <class RayT, class HitInfoT>
	void fill(const RayT &ray, const rtmi_hit &h, HitInfoT &o) const {
		TriT *t = tri_of_id_[h.tri];
		o.ray = ray;                    // object.cpp:15, scene.cpp:201
		o.normal = t->triNormal;        // :16 (face normal, not interpolated)
		o.distance = h.t;               // :17
		o.uv = t->textVert;             // :18
		o.BayCentricU = h.u;            // :19-20 (floats in the reference too)
		o.BayCentricV = h.v;
		o.tri = t;                      // :21
		o.object = object_of_tri_[h.tri]; // scene.cpp:200
		// :23  point = origin + (direction * ((triVert1 − origin) * triNormal)) / (triNormal * direction)
		const double ox = ray.origin.x, oy = ray.origin.y, oz = ray.origin.z;
		const double dx = ray.direction.x, dy = ray.direction.y, dz = ray.direction.z;
		const double nx = t->triNormal.x, ny = t->triNormal.y, nz = t->triNormal.z;
		const double s = ((t->triVert1.x - ox) * nx) + ((t->triVert1.y - oy) * ny) + ((t->triVert1.z - oz) * nz);
		const double den = (nx * dx) + (ny * dy) + (nz * dz);
		o.point.x = ox + (dx * s) / den;
		o.point.y = oy + (dy * s) / den;
		o.point.z = oz + (dz * s) / den;
	}

	rtmi_scene *scene_ = nullptr;
	std::vector<TriT *> tri_of_id_;
	std::vector<const ObjectT *> object_of_tri_;
};

// ------------------------------------------------------------------------------------------------ shading
// Scene::Trace / Shader::Shade / the raster loop on the device.  Materials are read from sceneObjects[k]->material
// (Render2/ray_tracer.h:23-34), lights from scene.GetLight(i) exactly as Shade reads them: Center(GetBox(*light)) and
// material->emission (shader.cpp:486-489) — GetBox and Center are the reference's own free functions (util.h), found
// at instantiation.  Scene::max_tree_depth is honoured; AO rays and radius are Shade's constants (shader.cpp:481).
template <class SceneT, class ObjectT, class TriT>
class ShaderBinding {
public:
	ShaderBinding() = default;
	ShaderBinding(const ShaderBinding &) = delete;
	ShaderBinding &operator=(const ShaderBinding &) = delete;
	~ShaderBinding() { rtmi_shader_destroy(shader_); }

	bool bind(const SceneBinding<SceneT, ObjectT, TriT> &geometry, const SceneT &scene, uint32_t ao_rays = 10, double ao_radius = 2.0,
	          uint64_t seed = 1) {
		std::vector<double> nrm(3 * geometry.triangles());
		for (size_t i = 0; i < geometry.triangles(); ++i) put3(&nrm[3 * i], geometry.triangle(i)->triNormal);
		std::vector<rtmi_material> mats;
		for (const ObjectT *o : scene.sceneObjects) {
			rtmi_material m;
			rgb(m.diffuse, o->material->diffuse); rgb(m.specular, o->material->specular); rgb(m.ambient, o->material->ambient);
			rgb(m.reflectivity, o->material->reflectivity); rgb(m.translucency, o->material->translucency);
			m.phong_exp = o->material->Phong_exp; m.ref_index = o->material->ref_index; m.roughness = o->material->microFacetsRoughness;
			mats.push_back(m);
		}
		std::vector<rtmi_light> lights;
		for (unsigned i = 0; i < scene.NumLights(); ++i) {
			rtmi_light l;
			put3(l.center, Center(GetBox(*scene.GetLight(i))));
			rgb(l.emission, scene.GetLight(i)->material->emission);
			lights.push_back(l);
		}
		params_.max_tree_depth = scene.max_tree_depth;
		params_.ao_rays = ao_rays; params_.ao_radius = ao_radius; params_.seed = seed;
		params_.flags = scene.radiosity != nullptr ? RTMI_SHADE_RADIOSITY_BRANCH : 0u; // which branch Shade takes, shader.cpp:431
		params_.reserved = 0;
		rtmi_shader_destroy(shader_);
		shader_ = rtmi_shader_create(geometry.handle(), nrm.data(), mats.data(), (uint32_t)mats.size(), lights.data(), (uint32_t)lights.size());
		return shader_ != nullptr;
	}

	// Color Scene::Trace(const Ray&) for a batch (rays[i].generation is honoured per batch: all rays of one call share
	// rays[0].generation).  ColorT needs red, green, blue.
	template <class RayT, class ColorT>
	void Trace(const RayT *rays, size_t n, ColorT *out) const {
		if (n == 0) return;
		std::vector<rtmi_ray> r(n);
		for (size_t i = 0; i < n; ++i) { put3(r[i].o, rays[i].origin); put3(r[i].d, rays[i].direction); r[i].tmax = RTMI_INFINITY; r[i].reserved = 0.0; }
		Buffer d_rays(n * sizeof(rtmi_ray)), d_rgb(3 * n * sizeof(double));
		if (hipMemcpy(d_rays.p, r.data(), n * sizeof(rtmi_ray), hipMemcpyHostToDevice) != hipSuccess) throw std::runtime_error("upload of the rays failed");
		if (rtmi_trace_dev(shader_, (const rtmi_ray *)d_rays.p, n, (uint32_t)rays[0].generation, 0, &params_, (double *)d_rgb.p, nullptr) != 0) throw Error("rtmi_trace_dev");
		std::vector<double> c(3 * n);
		if (hipMemcpy(c.data(), d_rgb.p, 3 * n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) throw std::runtime_error("download of the colours failed");
		for (size_t i = 0; i < n; ++i) { out[i].red = c[3 * i]; out[i].green = c[3 * i + 1]; out[i].blue = c[3 * i + 2]; }
	}

	// The loop of Rasterizer::Serial_Normal_Raster (rasterizer.cpp:219-233): I(i, j) = ToneMap(scene.Trace(ray(i, j))).
	// CameraT: Render2/ray_tracer.h:113-122; ImageT: PPM_Image (operator()(i, j) -> Pixel& with r, g, b).
	template <class CameraT, class ImageT>
	void Rasterize(const CameraT &cam, ImageT &I) const {
		rtmi_camera c;
		double eye[3], lookat[3], up[3];
		put3(eye, cam.eye); put3(lookat, cam.lookat); put3(up, cam.up);
		const double xw[2] = {cam.x_win.min, cam.x_win.max}, yw[2] = {cam.y_win.min, cam.y_win.max};
		if (rtmi_camera_setup(eye, lookat, up, cam.vpdist, xw, yw, cam.x_res, cam.y_res, &c) != 0) throw Error("rtmi_camera_setup");
		const size_t n = (size_t)cam.x_res * cam.y_res;
		Buffer d_pix(3 * n);
		if (rtmi_render_frame_dev(shader_, &c, 0.5, 0.5, 64, 64, 0, 1, &params_, nullptr, (uint8_t *)d_pix.p, nullptr) != 0) throw Error("rtmi_render_frame_dev");
		std::vector<uint8_t> px(3 * n);
		if (hipMemcpy(px.data(), d_pix.p, 3 * n, hipMemcpyDeviceToHost) != hipSuccess) throw std::runtime_error("download of the image failed");
		for (unsigned i = 0; i < cam.y_res; ++i)
			for (unsigned j = 0; j < cam.x_res; ++j) {
				const uint8_t *p = &px[3 * ((size_t)i * cam.x_res + j)];
				auto &dst = I(i, j);
				dst.r = p[0]; dst.g = p[1]; dst.b = p[2];
			}
	}

private:
	struct Buffer {
		void *p = nullptr;
		explicit Buffer(size_t bytes) { if (hipMalloc(&p, bytes) != hipSuccess) throw std::runtime_error("hipMalloc failed"); }
		~Buffer() { (void)hipFree(p); }
		Buffer(const Buffer &) = delete;
		Buffer &operator=(const Buffer &) = delete;
	};
	template <class ColorT> static void rgb(double *p, const ColorT &c) { p[0] = c.red; p[1] = c.green; p[2] = c.blue; }
	rtmi_shader *shader_ = nullptr;
	rtmi_trace_params params_ = {1, 10, 2.0, 1, 0, 0};
};

// ------------------------------------------------------------------------------------------------ radiosity
// The elements (radiosity.cpp:137-163: the BSP leaves, in element order) are both the patches and the occluders,
// so the binding builds its own scene from them.  ElemT needs triVert1..3, triNormal and radiosityValue
// (QuadTreeNode, ray_tracer.h:244-286).
template <class ElemT>
class RadiosityBinding {
public:
	RadiosityBinding() = default;
	RadiosityBinding(const RadiosityBinding &) = delete;
	RadiosityBinding &operator=(const RadiosityBinding &) = delete;
	~RadiosityBinding() {
		rtmi_patches_destroy(patches_);
		rtmi_scene_destroy(scene_);
		if (d_F_) (void)hipFree(d_F_);
	}

	// light_center: Center(GetBox(light)) as radiosity.cpp:270-272 computes it, or nullptr for no light
	bool setElements(const std::vector<ElemT *> &elements, const double *light_center) {
		elems_ = elements;
		const size_t ne = elems_.size();
		std::vector<double> xyz(9 * ne), nrm(3 * ne);
		for (size_t i = 0; i < ne; ++i) {
			put3(&xyz[9 * i], elems_[i]->triVert1); put3(&xyz[9 * i + 3], elems_[i]->triVert2); put3(&xyz[9 * i + 6], elems_[i]->triVert3);
			put3(&nrm[3 * i], elems_[i]->triNormal);
		}
		has_light_ = light_center != nullptr;
		if (has_light_) { light_[0] = light_center[0]; light_[1] = light_center[1]; light_[2] = light_center[2]; }
		n_ = (uint32_t)ne + (has_light_ ? 1u : 0u);
		ld_ = (n_ + 3u) / 4u * 4u;
		rtmi_patches_destroy(patches_); patches_ = nullptr;
		rtmi_scene_destroy(scene_);
		scene_ = rtmi_scene_create(xyz.data(), nullptr, ne);
		if (!scene_) return false;
		patches_ = rtmi_patches_create(scene_, xyz.data(), nrm.data());
		return patches_ != nullptr;
	}

	uint32_t n() const { return n_; } // elements + light, the order of the reference's FF / K / B (radiosity.cpp:8-14)

	// Radiosity::findFormFactor (radiosity.cpp:335-403): samples = 10, light_samples = 30 in the reference
	void findFormFactor(uint32_t samples = 10, uint32_t light_samples = 30, uint64_t seed = 1, uint32_t flags = 0) {
		if (d_F_) (void)hipFree(d_F_);
		d_F_ = nullptr;
		if (hipMalloc((void **)&d_F_, sizeof(float) * (size_t)n_ * ld_) != hipSuccess) throw std::runtime_error("hipMalloc(F) failed");
		if (rtmi_form_factors_dev(patches_, has_light_ ? light_ : nullptr, samples, light_samples, seed, flags, 0, n_, d_F_, ld_, nullptr) != 0)
			throw Error("rtmi_form_factors_dev");
		if (hipDeviceSynchronize() != hipSuccess) throw std::runtime_error("form-factor kernel failed");
	}

	// FF[i][j] as the reference's global holds it (row-major n × n, double)
	void downloadFF(double *FF) const {
		std::vector<float> f((size_t)n_ * ld_);
		if (hipMemcpy(f.data(), d_F_, f.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) throw std::runtime_error("download of F failed");
		for (uint32_t i = 0; i < n_; ++i)
			for (uint32_t j = 0; j < n_; ++j) FF[(size_t)i * n_ + j] = f[(size_t)i * ld_ + j];
	}

	// Radiosity::progressiveRefinement (radiosity.cpp:483-563): E = 0 except the light's 1370·50 (:499), ρ = 0.70
	// (base.h:42); returns B (n doubles, element order, light last) — the reference's global B.
	std::vector<double> progressiveRefinement(double rho = 0.70, double light_power = 1370.0 * 50.0, float tol_rel = 1e-7f,
	                                          rtmi_solve_stats *stats = nullptr) {
		std::vector<float> E(3 * (size_t)n_, 0.f), B(3 * (size_t)n_, 0.f);
		if (has_light_) E[3 * (n_ - 1)] = E[3 * (n_ - 1) + 1] = E[3 * (n_ - 1) + 2] = (float)light_power;
		float *d_E = nullptr, *d_B = nullptr;
		const size_t vb = sizeof(float) * 3 * (size_t)n_;
		if (hipMalloc((void **)&d_E, vb) != hipSuccess || hipMalloc((void **)&d_B, vb) != hipSuccess) throw std::runtime_error("hipMalloc(B) failed");
		(void)hipMemcpy(d_E, E.data(), vb, hipMemcpyHostToDevice);
		const double rho3[3] = {rho, rho, rho};
		rtmi_solve_stats st;
		const int rc = rtmi_radiosity_solve_dev(d_F_, ld_, n_, 0, n_, rho3, d_E, d_B, tol_rel, 10000, 4, nullptr, nullptr, &st, nullptr);
		if (rc == 0) (void)hipMemcpy(B.data(), d_B, vb, hipMemcpyDeviceToHost);
		(void)hipFree(d_E);
		(void)hipFree(d_B);
		if (rc != 0) throw Error("rtmi_radiosity_solve_dev");
		if (stats) *stats = st;
		B_.assign(n_, 0.0);
		for (uint32_t i = 0; i < n_; ++i) B_[i] = (double)B[3 * (size_t)i];
		return B_;
	}

	// Radiosity::setRadiosityForAllElements (radiosity.cpp:122-134)
	void setRadiosityForAllElements() {
		for (size_t i = 0; i < elems_.size(); ++i) elems_[i]->radiosityValue = B_[i];
	}

	// The pixel loop of Rasterizer::Radiosity_Raster (rasterizer.cpp:436-455):
	// I(i, j) = ToneMapRadiosity(radiosityHelper->trace_ray(ray(i, j), elements)).  The shared-vertex averages of
	// returnQuadTreeColor (radiosity_helper.h:156-213, over QuadTreeNode::nextAdj) stay the reference's: its own function is
	// evaluated at the three vertices of every element, where its barycentric weights are exactly (1,0,0), (0,1,0),
	// (0,0,1) — found at instantiation, so the host must have included radiosity_helper.h.  Reads radiosityValue.
	template <class CameraT, class ImageT>
	void Radiosity_Raster(const CameraT &cam, ImageT &I) const {
		const size_t ne = elems_.size();
		std::vector<double> vr(3 * ne);
		for (size_t i = 0; i < ne; ++i) {
			auto v1 = elems_[i]->triVert1, v2 = elems_[i]->triVert2, v3 = elems_[i]->triVert3;
			vr[3 * i] = returnQuadTreeColor(v1, elems_[i]).red;
			vr[3 * i + 1] = returnQuadTreeColor(v2, elems_[i]).red;
			vr[3 * i + 2] = returnQuadTreeColor(v3, elems_[i]).red;
		}
		rtmi_camera c;
		double eye[3], lookat[3], up[3];
		put3(eye, cam.eye); put3(lookat, cam.lookat); put3(up, cam.up);
		const double xw[2] = {cam.x_win.min, cam.x_win.max}, yw[2] = {cam.y_win.min, cam.y_win.max};
		if (rtmi_camera_setup(eye, lookat, up, cam.vpdist, xw, yw, cam.x_res, cam.y_res, &c) != 0) throw Error("rtmi_camera_setup");
		const size_t n = (size_t)cam.x_res * cam.y_res;
		double *d_vr = nullptr;
		uint8_t *d_pix = nullptr;
		if (hipMalloc((void **)&d_vr, sizeof(double) * 3 * ne) != hipSuccess || hipMalloc((void **)&d_pix, 3 * n) != hipSuccess) {
			if (d_vr) (void)hipFree(d_vr);
			throw std::runtime_error("hipMalloc failed");
		}
		std::vector<uint8_t> px(3 * n);
		bool ok = hipMemcpy(d_vr, vr.data(), sizeof(double) * 3 * ne, hipMemcpyHostToDevice) == hipSuccess;
		const int rc = ok ? rtmi_radiosity_raster_dev(patches_, d_vr, &c, 0.5, 0.5, 0, 0, 0, 1, nullptr, d_pix, nullptr) : -1;
		ok = ok && rc == 0 && hipMemcpy(px.data(), d_pix, 3 * n, hipMemcpyDeviceToHost) == hipSuccess;
		(void)hipFree(d_vr);
		(void)hipFree(d_pix);
		if (!ok) throw Error("rtmi_radiosity_raster_dev");
		for (unsigned i = 0; i < cam.y_res; ++i)
			for (unsigned j = 0; j < cam.x_res; ++j) {
				const uint8_t *p = &px[3 * ((size_t)i * cam.x_res + j)];
				auto &dst = I(i, j);
				dst.r = p[0]; dst.g = p[1]; dst.b = p[2];
			}
	}

private:
	std::vector<ElemT *> elems_;
	std::vector<double> B_;
	rtmi_scene *scene_ = nullptr;
	rtmi_patches *patches_ = nullptr;
	float *d_F_ = nullptr;
	double light_[3] = {0, 0, 0};
	bool has_light_ = false;
	uint32_t n_ = 0;
	uint64_t ld_ = 0;
};

} // namespace rtmi_host
#endif
