// librtmi — host-side reader of the RayTracer's scene files (.sdf + .obj): flat arrays for rtmi_scene_create /
// rtmi_shader_create / rtmi_camera_setup, so a scene of the reference renders without its host driver.
// Restates Scene::BuildScene (Render2/scene.cpp:209-460) and ParamReader (Render2/params.cpp), quirks included,
// through the same C library calls (sscanf, strtok, atoi, istream::getline with a 512-byte buffer):
//   - a key matches as a PREFIX at the start of the line (params.cpp:11-20), so "vt" and "vn" are tried before "v"
//     (scene.cpp:412-429) and every command is tried in the reference's order (scene.cpp:284-300);
//   - a line of 512 bytes or more ends the file (scene.cpp:211, 262, 393: getline fails);
//   - the material is reset after every `object`, never after a light (scene.cpp:337-351);
//   - an .obj must start with `o <name of the object>` (scene.cpp:400-405), faces are triangles or quads
//     (params.cpp:214: a quad becomes (v1,v2,v3) + (v3,v4,v1)); after the first token a face line is split at
//     " ,.-" (params.cpp:146-148);
//   - triNormal = Unit(mean of the face's vertex normals) when the face has them, else Unit((v2−v1)×(v3−v2));
//     the second triangle of a quad always gets the geometric normal, flipped toward its vertex normal
//     (params.cpp:188-235);
//   - flipYZ and translate are applied to the vertices as they are read (scene.cpp:424-432); rotation and scale are
//     parsed and ignored, as in the reference;
//   - lights: `spherelight (x, y, z) r` (spherelight.cpp:16-25) with the material current at that line; a light counts
//     if its emission is not black (scene.cpp:276); its position for shading is Center(GetBox(light))
//     (aabb.h:31-33, spherelight.cpp:27-32).  No scene of the reference uses `quadlight`; it is rejected here.
// Pure host code, no GPU.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "../../include/rtmi.h"
#include "rtmi_host.hpp"

namespace {

struct V3 { double x = 0, y = 0, z = 0; };
V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
V3 operator/(V3 a, double c) { return {a.x / c, a.y / c, a.z / c}; }
double dot(V3 a, V3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
V3 unit(V3 a) { // vec3.h:98
	const double d = a.x * a.x + a.y * a.y + a.z * a.z;
	return d > 0.0 ? a / std::sqrt(d) : V3{};
}

struct Col { double r = 0, g = 0, b = 0; };
const Col kWhite = {1.0, 1.0, 1.0};
struct Mat { Col diffuse, specular, emission, ambient, reflectivity, translucency; double rough = 0, phong = 0, ref_index = 0; };

// Render2/params.cpp, function by function
class ParamReader {
public:
	explicit ParamReader(const std::string &s) : params(s) {}
	bool key(const std::string &field) { // operator[](string)
		SkipBlanks();
		if (params.find(field, 0) == 0) { params.erase(0, field.length()); return true; }
		return false;
	}
	bool getString(std::string &field) {
		SkipBlanks();
		char c[512];
		if (sscanf(params.c_str(), "%511s", c) == 1) { field = c; params.erase(0, field.length()); return true; }
		return false;
	}
	bool getPoint(V3 &v) {
		SkipBlanks();
		return sscanf(params.c_str(), "%lf %lf %lf", &v.x, &v.y, &v.z) == 3;
	}
	bool interval(double &lo, double &hi) {
		if (sscanf(params.c_str(), " ( %lf , %lf )", &lo, &hi) == 2) { params.erase(0, 1 + params.find(")", 0)); return true; }
		return false;
	}
	bool color(Col &c) {
		if (sscanf(params.c_str(), " [ %lf , %lf , %lf ]", &c.r, &c.g, &c.b) == 3) { params.erase(0, 1 + params.find("]", 0)); return true; }
		return false;
	}
	bool vec3(V3 &v) {
		SkipBlanks();
		if (sscanf(params.c_str(), "(%lf , %lf , %lf)", &v.x, &v.y, &v.z) == 3) { params.erase(0, 1 + params.find(")", 0)); return true; }
		return false;
	}
	bool real(double &x) {
		SkipBlanks();
		if (sscanf(params.c_str(), "%lf", &x) == 1) {
			const int len = 1 + (int)params.find(" ", 0);
			if (len > 1) params.erase(0, len);
			return true;
		}
		return false;
	}
	bool uns(unsigned &x) {
		SkipBlanks();
		if (sscanf(params.c_str(), "%u", &x) == 1) {
			const int len = 1 + (int)params.find(" ", 0);
			if (len > 1) params.erase(0, len);
			return true;
		}
		return false;
	}
	std::string rest() { SkipBlanks(); return params; }

private:
	void SkipBlanks() { while (!params.empty() && (params[0] == ' ' || params[0] == '\t')) params.erase(0, 1); }
	std::string params;
};

bool Skip(const char *line) { // scene.cpp:29-36
	for (char c = *line; c != '\0' && c != '\n' && c != '#'; c = *++line)
		if (c != ' ' && c != '\t') return false;
	return true;
}

struct Obj {
	std::string name;
	Mat material;
	bool flipYZ = false, translate = false;
	V3 translation;
	std::vector<V3> points, normals;
	std::vector<double> tri_xyz, tri_nrm; // 9 / 3 per triangle
};

struct Light { V3 center; double radius; Col emission; };

// ParamReader::returnCoordOfFace, params.cpp:132-245; `line` = what follows the "f"
bool read_face(Obj &o, std::string line, std::string &err) {
	while (!line.empty() && (line[0] == ' ' || line[0] == '\t')) line.erase(0, 1);
	std::vector<char> w(line.begin(), line.end());
	w.push_back('\0');
	std::vector<char *> tok;
	for (char *p = strtok(w.data(), " "); p != NULL; p = strtok(NULL, " ,.-")) tok.push_back(p);
	if (tok.size() < 3) { err = "face with fewer than 3 vertices"; return false; }
	int fields = 1;
	bool has_vn = false;
	for (int x = 0; tok[0][x] != '\0'; ++x)
		if (tok[0][x] == '/') {
			if (tok[0][x + 1] == '/') { fields--; has_vn = true; }
			fields++;
		}
	if (fields == 3) has_vn = true;
	std::vector<std::vector<int>> fd(tok.size());
	for (size_t i = 0; i < tok.size(); ++i)
		for (int j = 0; j <= fields; ++j) {
			char *p = (j == 0) ? strtok(tok[i], "/") : strtok(NULL, "/");
			if (p != NULL) fd[i].push_back(atoi(p) - 1);
		}
	auto pt = [&](size_t i, V3 &out) {
		if (fd[i].empty() || fd[i][0] < 0 || (size_t)fd[i][0] >= o.points.size()) return false;
		out = o.points[fd[i][0]];
		return true;
	};
	auto vn = [&](size_t i, V3 &out) {
		if ((int)fd[i].size() < fields || fd[i][fields - 1] < 0 || (size_t)fd[i][fields - 1] >= o.normals.size()) return false;
		out = o.normals[fd[i][fields - 1]];
		return true;
	};
	auto emit = [&](V3 a, V3 b, V3 c, V3 n) {
		const double v[9] = {a.x, a.y, a.z, b.x, b.y, b.z, c.x, c.y, c.z};
		o.tri_xyz.insert(o.tri_xyz.end(), v, v + 9);
		o.tri_nrm.push_back(n.x); o.tri_nrm.push_back(n.y); o.tri_nrm.push_back(n.z);
	};
	V3 p1, p2, p3;
	if (!pt(0, p1) || !pt(1, p2) || !pt(2, p3)) { err = "face refers to a vertex that does not exist"; return false; }
	V3 tri_normal;
	if (has_vn) { // params.cpp:190-195
		V3 n1, n2, n3;
		if (!vn(0, n1) || !vn(1, n2) || !vn(2, n3)) { err = "face refers to a vertex normal that does not exist"; return false; }
		tri_normal = unit(((n1 + n2) + n3) / 3);
	} else {
		tri_normal = unit(cross(p2 - p1, p3 - p2)); // :199-208
	}
	emit(p1, p2, p3, tri_normal);
	if (tok.size() == 4) { // :214-241
		V3 p4;
		if (!pt(3, p4)) { err = "face refers to a vertex that does not exist"; return false; }
		V3 calc = unit(cross(p4 - p3, p1 - p4));
		if (has_vn) {
			V3 n3;
			if (!vn(2, n3)) { err = "face refers to a vertex normal that does not exist"; return false; }
			if (dot(calc, n3) < 0.0) calc = -1.0 * calc;
		}
		emit(p3, p4, p1, calc);
	}
	return true;
}

} // namespace

struct rtmi_scene_file {
	std::vector<Obj> objects;
	std::vector<Light> lights;
	double cam14[14];
	unsigned res[2];
	uint64_t n_tris = 0;
};

using rtmi::fail;

extern "C" {

rtmi_scene_file *rtmi_scene_file_load(const char *scenes_dir, const char *stem) {
	const char *who = "rtmi_scene_file_load";
	if (!scenes_dir || !stem) { fail(std::string(who) + ": NULL argument"); return nullptr; }
	std::string dir = scenes_dir;
	if (!dir.empty() && dir.back() != '/') dir += '/';
	const std::string sdf = dir + stem + ".sdf";
	std::ifstream fin(sdf.c_str());
	if (fin.fail()) { fail(std::string(who) + ": could not open " + sdf); return nullptr; }
	rtmi_scene_file *sf = new rtmi_scene_file();
	V3 eye, lookat, up;
	double vpdist = 0.0, xw[2] = {-1.0, 1.0}, yw[2] = {-1.0, 1.0};
	sf->res[0] = 400; sf->res[1] = 400; // base.h:68-69
	Mat material;
	material.diffuse = kWhite; material.specular = kWhite; // scene.cpp:238-246
	bool flip = false, translate = false;
	V3 translation, ignored;
	char input_line[512];
	while (fin.getline(input_line, 512)) { // scene.cpp:262-355
		if (Skip(input_line)) continue;
		{ // SphereLight::ReadString, spherelight.cpp:16-25
			ParamReader get(input_line);
			V3 c;
			double r;
			if (get.key("spherelight") && get.vec3(c) && get.real(r)) {
				if (material.emission.r != 0.0 || material.emission.g != 0.0 || material.emission.b != 0.0) // scene.cpp:276
					sf->lights.push_back({c, r, material.emission});
				continue;
			}
		}
		{
			ParamReader q(input_line);
			if (q.key("quadlight")) { delete sf; fail(std::string(who) + ": quadlight is not supported (no scene of the reference uses it)"); return nullptr; }
		}
		ParamReader get(input_line);
		unsigned u;
		if (get.key("eye") && get.vec3(eye)) continue;
		if (get.key("lookat") && get.vec3(lookat)) continue;
		if (get.key("up") && get.vec3(up)) continue;
		if (get.key("vpdist") && get.real(vpdist)) continue;
		if (get.key("x_res") && get.uns(u)) { sf->res[0] = u; continue; }
		if (get.key("y_res") && get.uns(u)) { sf->res[1] = u; continue; }
		if (get.key("x_win") && get.interval(xw[0], xw[1])) continue;
		if (get.key("y_win") && get.interval(yw[0], yw[1])) continue;
		if (get.key("ambient") && get.color(material.ambient)) continue;
		if (get.key("diffuse") && get.color(material.diffuse)) continue;
		if (get.key("specular") && get.color(material.specular)) continue;
		if (get.key("emission") && get.color(material.emission)) continue;
		if (get.key("reflectivity") && get.color(material.reflectivity)) continue;
		if (get.key("translucency") && get.color(material.translucency)) continue;
		if (get.key("Phong_exp") && get.real(material.phong)) continue;
		if (get.key("ref_index") && get.real(material.ref_index)) continue;
		if (get.key("roughness") && get.real(material.rough)) continue;
		std::string word;
		if (get.key("flipYZ") && get.getString(word)) {
			if (word == "0" || word == "1") flip = word == "1";
			continue;
		}
		if (get.key("translate") && get.vec3(translation)) { translate = true; continue; }
		if (get.key("rotation") && get.vec3(ignored)) continue; // parsed, never applied (scene.cpp:313-316)
		if (get.key("scale") && get.vec3(ignored)) continue;
		if (get.key("object") && get.getString(word)) { // scene.cpp:322-352
			Obj o;
			o.name = word; o.material = material; o.flipYZ = flip; o.translate = translate; o.translation = translation;
			sf->objects.push_back(o);
			material = Mat();
			material.diffuse = kWhite; material.specular = kWhite;
			flip = false; translate = false; translation = V3();
			continue;
		}
		// anything else: the reference prints a warning unless it is "end" and goes on (scene.cpp:356-361)
	}
	// ---- object files, scene.cpp:366-450
	for (Obj &o : sf->objects) {
		const std::string path = dir + o.name + ".obj";
		std::ifstream fo(path.c_str());
		if (fo.fail()) continue; // "Could not open file": the object stays empty (scene.cpp:380-383)
		int line_num = 0;
		while (fo.getline(input_line, 512)) {
			if (Skip(input_line)) continue;
			ParamReader get(input_line);
			if (line_num == 0) {
				std::string name;
				if (get.key("o") && get.getString(name) && name == o.name) { line_num++; continue; }
				break;
			}
			V3 v;
			if (get.key("mtllib")) {
			} else if (get.key("vt")) {
			} else if (get.key("vn")) {
				if (get.getPoint(v)) o.normals.push_back(v);
			} else if (get.key("v")) {
				if (get.getPoint(v)) {
					V3 p = o.flipYZ ? V3{v.x, v.z, v.y} : v; // scene.cpp:424-425
					if (o.translate) { p.x += o.translation.x; p.y += o.translation.y; p.z += o.translation.z; }
					o.points.push_back(p);
				}
			} else if (get.key("f")) {
				std::string err;
				if (!read_face(o, get.rest(), err)) {
					const std::string msg = std::string(who) + ": " + path + ": " + err + ": " + input_line;
					delete sf;
					fail(msg);
					return nullptr;
				}
			}
			line_num++;
		}
		sf->n_tris += o.tri_nrm.size() / 3;
	}
	const double c14[14] = {eye.x, eye.y, eye.z, lookat.x, lookat.y, lookat.z, up.x, up.y, up.z, vpdist, xw[0], xw[1], yw[0], yw[1]};
	std::memcpy(sf->cam14, c14, sizeof c14);
	return sf;
}

void rtmi_scene_file_free(rtmi_scene_file *sf) { delete sf; }

int rtmi_scene_file_info(const rtmi_scene_file *sf, uint64_t *n_tris, uint32_t *n_objects, uint32_t *n_lights) {
	if (!sf) return fail("rtmi_scene_file_info: NULL scene file");
	if (n_tris) *n_tris = sf->n_tris;
	if (n_objects) *n_objects = (uint32_t)sf->objects.size();
	if (n_lights) *n_lights = (uint32_t)sf->lights.size();
	return 0;
}

int rtmi_scene_file_get(const rtmi_scene_file *sf, double *tri_xyz, uint32_t *tri_object, double *tri_normals, rtmi_material *materials,
                        rtmi_light *lights, double cam14[14], uint32_t res[2]) {
	if (!sf) return fail("rtmi_scene_file_get: NULL scene file");
	uint64_t at = 0;
	uint32_t k = 0;
	for (const Obj &o : sf->objects) {
		const uint64_t n = o.tri_nrm.size() / 3;
		if (tri_xyz && n) std::memcpy(tri_xyz + 9 * at, o.tri_xyz.data(), sizeof(double) * 9 * n);
		if (tri_normals && n) std::memcpy(tri_normals + 3 * at, o.tri_nrm.data(), sizeof(double) * 3 * n);
		if (tri_object) for (uint64_t i = 0; i < n; ++i) tri_object[at + i] = k;
		if (materials) {
			const Mat &m = o.material;
			rtmi_material &d = materials[k];
			const Col *src[5] = {&m.diffuse, &m.specular, &m.ambient, &m.reflectivity, &m.translucency};
			double *dst[5] = {d.diffuse, d.specular, d.ambient, d.reflectivity, d.translucency};
			for (int a = 0; a < 5; ++a) { dst[a][0] = src[a]->r; dst[a][1] = src[a]->g; dst[a][2] = src[a]->b; }
			d.phong_exp = m.phong; d.ref_index = m.ref_index; d.roughness = m.rough;
		}
		at += n;
		++k;
	}
	if (lights)
		for (size_t i = 0; i < sf->lights.size(); ++i) {
			const Light &L = sf->lights[i];
			const double c[3] = {L.center.x, L.center.y, L.center.z};
			for (int a = 0; a < 3; ++a) {
				// GetSlab along a unit axis (spherelight.cpp:27-32): dot = (v·center)/1, [dot − r, dot + r]/1; Center = 0.5·(min + max)
				const double e[3] = {a == 0 ? 1.0 : 0.0, a == 1 ? 1.0 : 0.0, a == 2 ? 1.0 : 0.0};
				const double d = ((e[0] * c[0]) + (e[1] * c[1]) + (e[2] * c[2])) / 1.0;
				lights[i].center[a] = 0.5 * (((d - L.radius) / 1.0) + ((d + L.radius) / 1.0));
			}
			lights[i].emission[0] = L.emission.r; lights[i].emission[1] = L.emission.g; lights[i].emission[2] = L.emission.b;
		}
	if (cam14) std::memcpy(cam14, sf->cam14, sizeof sf->cam14);
	if (res) { res[0] = sf->res[0]; res[1] = sf->res[1]; }
	return 0;
}

} // extern "C"
