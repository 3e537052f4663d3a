// librtmi — host-side reader of the RayTracer's scene files (.sdf + .obj): flat arrays for rtmi_scene_create /
// rtmi_shader_create / rtmi_camera_setup, so a scene of the reference renders without its host driver.
// Reads what Scene::BuildScene (Render2/scene.cpp:209-460) reads, with the same result on the reference's scene files
// (every triangle, normal, material, light and camera value of eight scenes equals the compiled reference's own,
// tests/golden/g7_scene_files.npz) — its quirks included in the default, STRICT mode:
//   - a key matches as a PREFIX at the start of the line, so "vt" and "vn" are tried before "v" and every command in the
//     reference's order (scene.cpp:284-300, 412-429);
//   - a line of 512 bytes or more ends the file (scene.cpp:211, 262, 393);
//   - the material is reset after every `object`, never after a light (scene.cpp:337-351);
//   - an .obj must start with `o <name of the object>` (scene.cpp:400-405); faces are triangles or quads (params.cpp:214:
//     a quad becomes (v1,v2,v3) + (v3,v4,v1)); after the first corner a face line is split at " ,.-" (params.cpp:146-148);
//   - triNormal = Unit(mean of the face's vertex normals) when the face has them, else Unit((v2−v1)×(v3−v2)); the second
//     triangle of a quad always gets the geometric normal, turned toward its vertex normal (params.cpp:188-235);
//   - flipYZ and translate are applied to the vertices as they are read (scene.cpp:424-432); rotation and scale are read
//     and ignored, as in the reference;
//   - lights: `spherelight (x, y, z) r` (spherelight.cpp:16-25) with the material current at that line; a light counts
//     if its emission is not black (scene.cpp:276); its position for shading is Center(GetBox(light))
//     (aabb.h:31-33, spherelight.cpp:27-32).  No scene of the reference uses `quadlight`; it is rejected here.
// RTMI_SCENE_LENIENT lifts the limits that keep the reference's larger assets from loading: lines of
// any length, faces with any number of corners (a fan around the first), any `o`/`g` name or none, relative indices.
// Pure host code, no GPU.
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
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

// ---- reading one line --------------------------------------------------------------------------------------------
// A cursor over the text of a line.  What each read accepts is what the reference's ParamReader accepts (Render2/params.cpp:
// its scanf formats let white space stand anywhere between the pieces of "(x, y, z)", "[r, g, b]" and "(lo, hi)", a key
// matches as a prefix, and after a bracketed value the rest of the line starts behind the FIRST closing bracket) — the
// fixtures of tests/golden/g7_scene_files.npz, made by the compiled reference's own BuildScene, pin that behaviour.
class Cursor {
public:
	explicit Cursor(const char *text) : at_(text) {}
	// the key, as a prefix of what is left (after blanks)
	bool word(const char *key) {
		blanks();
		const size_t n = std::strlen(key);
		if (std::strncmp(at_, key, n) != 0) return false;
		at_ += n;
		return true;
	}
	// the next run of non-space characters
	bool token(std::string &out) {
		blanks();
		const char *q = at_;
		while (std::isspace((unsigned char)*q)) ++q;
		const char *e = q;
		while (*e != '\0' && !std::isspace((unsigned char)*e) && e - q < 511) ++e;
		if (e == q) return false;
		out.assign(q, e);
		at_ += out.size(); // (the reference drops as many characters as the token has, from where the blanks ended)
		return true;
	}
	bool three_numbers(V3 &v) {
		blanks();
		const char *q = at_;
		return number(q, v.x) && number(q, v.y) && number(q, v.z);
	}
	// "(x, y, z)" — blanks first, then the bracket must be the very next character
	bool vector(V3 &v) {
		blanks();
		const char *q = at_;
		if (*q != '(') return false;
		++q;
		if (!(number(q, v.x) && mark(q, ',') && number(q, v.y) && mark(q, ',') && number(q, v.z))) return false;
		behind_first(')');
		return true;
	}
	// "(lo, hi)" and "[r, g, b]" — white space may come before the bracket
	bool pair(double &lo, double &hi) {
		const char *q = at_;
		if (!(mark(q, '(') && number(q, lo) && mark(q, ',') && number(q, hi))) return false;
		behind_first(')');
		return true;
	}
	bool triple(Col &c) {
		const char *q = at_;
		if (!(mark(q, '[') && number(q, c.r) && mark(q, ',') && number(q, c.g) && mark(q, ',') && number(q, c.b))) return false;
		behind_first(']');
		return true;
	}
	// a number; what is left then starts behind the first blank of the line's remainder (if there is one past its start)
	bool real(double &x) {
		blanks();
		const char *q = at_;
		if (!number(q, x)) return false;
		behind_first_space();
		return true;
	}
	bool count(unsigned &x) {
		blanks();
		const char *q = at_;
		while (std::isspace((unsigned char)*q)) ++q;
		char *e = nullptr;
		const unsigned long v = std::strtoul(q, &e, 10);
		if (e == q) return false;
		x = (unsigned)v;
		behind_first_space();
		return true;
	}
	std::string rest() { blanks(); return std::string(at_); }

private:
	void blanks() { while (*at_ == ' ' || *at_ == '\t') ++at_; }
	static bool number(const char *&q, double &x) { // strtod reads what scanf's %lf reads, leading white space included
		char *e = nullptr;
		const double v = std::strtod(q, &e);
		if (e == q) return false;
		x = v;
		q = e;
		return true;
	}
	static bool mark(const char *&q, char c) {
		while (std::isspace((unsigned char)*q)) ++q;
		if (*q != c) return false;
		++q;
		return true;
	}
	void behind_first(char c) {
		const char *q = std::strchr(at_, c);
		if (q) at_ = q + 1;
	}
	void behind_first_space() {
		const char *q = std::strchr(at_, ' ');
		if (q && q > at_) at_ = q + 1;
	}
	const char *at_;
};

bool blank_or_comment(const char *line) { // scene.cpp:29-36: nothing but blanks before the end, a newline or '#'
	for (; *line != '\0' && *line != '\n' && *line != '#'; ++line)
		if (*line != ' ' && *line != '\t') return false;
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

// ---- one face --------------------------------------------------------------------------------------------------
// Corners of a face line.  STRICT (the reference, params.cpp:132-245): the first corner ends at a blank, every later one at
// any of " ,.-"; a corner's fields are its non-empty pieces between '/'; the corner layout (how many fields, whether the
// last one is a normal) is read off the FIRST corner: "a", "a/b", "a//c", "a/b/c".  Indices are decimal prefixes, 1-based.
// LENIENT: corners are separated by blanks only, so negative (relative) indices survive.
struct Corner { int v = -1, n = -1; };

static int leading_int(const std::string &t) { return (int)std::strtol(t.c_str(), nullptr, 10); }

static std::vector<std::string> pieces(const std::string &text, const char *first_seps, const char *later_seps) {
	std::vector<std::string> out;
	size_t i = 0;
	const char *seps = first_seps;
	while (i < text.size()) {
		while (i < text.size() && std::strchr(seps, text[i])) ++i;
		size_t j = i;
		while (j < text.size() && !std::strchr(seps, text[j])) ++j;
		if (j > i) out.push_back(text.substr(i, j - i));
		i = j;
		seps = later_seps;
	}
	return out;
}

bool read_face(Obj &o, const std::string &line, bool lenient, std::string &err) {
	const std::vector<std::string> corners = lenient ? pieces(line, " \t\r", " \t\r") : pieces(line, " ", " ,.-");
	if (corners.size() < 3) { err = "face with fewer than 3 vertices"; return false; }
	// layout, from the first corner
	const std::string &c0 = corners[0];
	const size_t slashes = (size_t)std::count(c0.begin(), c0.end(), '/');
	const bool doubled = c0.find("//") != std::string::npos;
	const int fields = doubled ? (int)slashes : (int)slashes + 1; // "a//c" has two fields, "a/b/c" three
	const bool has_vn = doubled || fields == 3;
	std::vector<Corner> cs(corners.size());
	for (size_t k = 0; k < corners.size(); ++k) {
		const std::vector<std::string> f = pieces(corners[k], "/", "/");
		auto index_of = [&](const std::string &t, size_t count) {
			const int i = leading_int(t);
			return (lenient && i < 0) ? (int)count + i : i - 1; // relative indices in lenient mode
		};
		if (!f.empty()) cs[k].v = index_of(f[0], o.points.size());
		if (has_vn && (int)f.size() >= fields) cs[k].n = index_of(f[(size_t)fields - 1], o.normals.size());
	}
	auto point = [&](size_t k, V3 &out) {
		if (cs[k].v < 0 || (size_t)cs[k].v >= o.points.size()) { err = "face refers to a vertex that does not exist"; return false; }
		out = o.points[(size_t)cs[k].v];
		return true;
	};
	auto normal = [&](size_t k, V3 &out) {
		if (cs[k].n < 0 || (size_t)cs[k].n >= o.normals.size()) { err = "face refers to a vertex normal that does not exist"; return false; }
		out = o.normals[(size_t)cs[k].n];
		return true;
	};
	auto emit = [&](V3 a, V3 b, V3 c, V3 n) {
		const double v[9] = {a.x, a.y, a.z, b.x, b.y, b.z, c.x, c.y, c.z};
		o.tri_xyz.insert(o.tri_xyz.end(), v, v + 9);
		o.tri_nrm.push_back(n.x); o.tri_nrm.push_back(n.y); o.tri_nrm.push_back(n.z);
	};
	V3 p1, p2, p3;
	if (!point(0, p1) || !point(1, p2) || !point(2, p3)) return false;
	// first triangle: Unit(mean of its vertex normals) when the face has them, else Unit((v2 − v1) × (v3 − v2)) (params.cpp:188-208)
	V3 first_normal;
	if (has_vn) {
		V3 n1, n2, n3;
		if (!normal(0, n1) || !normal(1, n2) || !normal(2, n3)) return false;
		first_normal = unit(((n1 + n2) + n3) / 3);
	} else {
		first_normal = unit(cross(p2 - p1, p3 - p2));
	}
	emit(p1, p2, p3, first_normal);
	// A fourth corner closes a quad with (v3, v4, v1): geometric normal, turned toward v3's vertex normal (params.cpp:214-241).
	// The reference stops there; in lenient mode every further corner adds the fan triangle (v_{k-1}, v_k, v1) the same way.
	const size_t last = lenient ? corners.size() : std::min<size_t>(corners.size(), 4);
	if (!lenient && corners.size() != 4) return true; // strict: only exactly four corners make a second triangle
	V3 prev = p3;
	size_t prev_k = 2;
	for (size_t k = 3; k < last; ++k) {
		V3 pk;
		if (!point(k, pk)) return false;
		V3 g = unit(cross(pk - prev, p1 - pk));
		if (has_vn) {
			V3 nprev;
			if (!normal(prev_k, nprev)) return false;
			if (dot(g, nprev) < 0.0) g = -1.0 * g;
		}
		emit(prev, pk, p1, g);
		prev = pk;
		prev_k = k;
	}
	return true;
}

// one line of a file: the reference reads into a 512-byte buffer and stops the file at the first line that does not fit
// (scene.cpp:211, 262, 393); lenient mode takes lines of any length
static bool next_line(std::ifstream &f, std::string &line, bool lenient) {
	if (lenient) {
		if (!std::getline(f, line)) return false;
		if (!line.empty() && line.back() == '\r') line.pop_back();
		return true;
	}
	char buf[512];
	if (!f.getline(buf, sizeof buf)) return false;
	line = buf;
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

static rtmi_scene_file *scene_file_load(const char *scenes_dir, const char *stem, uint32_t flags, const char *who) {
	if (!scenes_dir || !stem) { fail(std::string(who) + ": NULL argument"); return nullptr; }
	const bool lenient = (flags & RTMI_SCENE_LENIENT) != 0;
	std::string dir = scenes_dir;
	if (!dir.empty() && dir.back() != '/') dir += '/';
	const std::string sdf = dir + stem + ".sdf";
	std::ifstream fin(sdf.c_str());
	if (fin.fail()) { fail(std::string(who) + ": could not open " + sdf); return nullptr; }
	std::unique_ptr<rtmi_scene_file> sf(new rtmi_scene_file());
	V3 eye, lookat, up;
	double vpdist = 0.0, xw[2] = {-1.0, 1.0}, yw[2] = {-1.0, 1.0};
	sf->res[0] = 400; sf->res[1] = 400; // base.h:68-69
	auto fresh_material = [] { // scene.cpp:238-246
		Mat m;
		m.diffuse = kWhite; m.specular = kWhite;
		return m;
	};
	Mat material = fresh_material();
	bool flip = false, translate = false;
	V3 translation, ignored;
	std::string text;
	// ---- the scene description: one command per line, tried in the reference's order (scene.cpp:262-361)
	while (next_line(fin, text, lenient)) {
		const char *line = text.c_str();
		if (blank_or_comment(line)) continue;
		{ // a light: "spherelight (x, y, z) r" with the material current at this line (spherelight.cpp:16-25); black ones do not count (scene.cpp:276)
			Cursor c(line);
			V3 centre;
			double radius;
			if (c.word("spherelight") && c.vector(centre) && c.real(radius)) {
				const Col &e = material.emission;
				if (e.r != 0.0 || e.g != 0.0 || e.b != 0.0) sf->lights.push_back({centre, radius, e});
				continue;
			}
		}
		if (Cursor(line).word("quadlight")) { fail(std::string(who) + ": quadlight is not supported (no scene of the reference uses it)"); return nullptr; }
		Cursor c(line);
		unsigned u;
		std::string word;
		if (c.word("eye") && c.vector(eye)) continue;
		if (c.word("lookat") && c.vector(lookat)) continue;
		if (c.word("up") && c.vector(up)) continue;
		if (c.word("vpdist") && c.real(vpdist)) continue;
		if (c.word("x_res") && c.count(u)) { sf->res[0] = u; continue; }
		if (c.word("y_res") && c.count(u)) { sf->res[1] = u; continue; }
		if (c.word("x_win") && c.pair(xw[0], xw[1])) continue;
		if (c.word("y_win") && c.pair(yw[0], yw[1])) continue;
		if (c.word("ambient") && c.triple(material.ambient)) continue;
		if (c.word("diffuse") && c.triple(material.diffuse)) continue;
		if (c.word("specular") && c.triple(material.specular)) continue;
		if (c.word("emission") && c.triple(material.emission)) continue;
		if (c.word("reflectivity") && c.triple(material.reflectivity)) continue;
		if (c.word("translucency") && c.triple(material.translucency)) continue;
		if (c.word("Phong_exp") && c.real(material.phong)) continue;
		if (c.word("ref_index") && c.real(material.ref_index)) continue;
		if (c.word("roughness") && c.real(material.rough)) continue;
		if (c.word("flipYZ") && c.token(word)) {
			if (word == "0" || word == "1") flip = word == "1";
			continue;
		}
		if (c.word("translate") && c.vector(translation)) { translate = true; continue; }
		if (c.word("rotation") && c.vector(ignored)) continue; // read, never applied (scene.cpp:313-316)
		if (c.word("scale") && c.vector(ignored)) continue;
		if (c.word("object") && c.token(word)) { // the object takes the material and transform set so far; both start afresh (scene.cpp:322-352)
			Obj o;
			o.name = word; o.material = material; o.flipYZ = flip; o.translate = translate; o.translation = translation;
			sf->objects.push_back(o);
			material = fresh_material();
			flip = false; translate = false; translation = V3();
			continue;
		}
		// anything else: the reference prints a warning unless it is "end", and goes on (scene.cpp:356-361)
	}
	// ---- the objects' .obj files (scene.cpp:366-450)
	for (Obj &o : sf->objects) {
		const std::string path = dir + o.name + ".obj";
		std::ifstream fo(path.c_str());
		if (fo.fail()) { // the reference says "Could not open file" and leaves the object empty (scene.cpp:380-383)
			if (lenient) { fail(std::string(who) + ": could not open " + path); return nullptr; }
			continue;
		}
		bool first = true;
		while (next_line(fo, text, lenient)) {
			const char *line = text.c_str();
			if (blank_or_comment(line)) continue;
			Cursor c(line);
			if (first) { // the reference insists on "o <the object's name>" as the first line and drops the file otherwise (scene.cpp:400-405)
				first = false;
				std::string name;
				if (!lenient) {
					if (c.word("o") && c.token(name) && name == o.name) continue;
					break;
				}
				if (Cursor(line).word("o ") || Cursor(line).word("g ")) continue; // lenient: any name, or no such line at all
			}
			V3 v;
			// a key is a prefix: "vt" and "vn" must be tried before "v" (scene.cpp:412-429)
			if (c.word("mtllib") || c.word("usemtl") || c.word("vt")) continue;
			if (lenient && (c.word("o ") || c.word("g ") || c.word("s "))) continue;
			if (c.word("vn")) {
				if (c.three_numbers(v)) o.normals.push_back(v);
			} else if (c.word("v")) {
				if (c.three_numbers(v)) {
					V3 p = o.flipYZ ? V3{v.x, v.z, v.y} : v; // scene.cpp:424-425
					if (o.translate) { p.x += o.translation.x; p.y += o.translation.y; p.z += o.translation.z; }
					o.points.push_back(p);
				}
			} else if (c.word("f")) {
				std::string err;
				if (!read_face(o, c.rest(), lenient, err)) { fail(std::string(who) + ": " + path + ": " + err + ": " + text.substr(0, 200)); return nullptr; }
			}
		}
		sf->n_tris += o.tri_nrm.size() / 3;
	}
	const double c14[14] = {eye.x, eye.y, eye.z, lookat.x, lookat.y, lookat.z, up.x, up.y, up.z, vpdist, xw[0], xw[1], yw[0], yw[1]};
	std::memcpy(sf->cam14, c14, sizeof c14);
	return sf.release();
}

rtmi_scene_file *rtmi_scene_file_load_ex(const char *scenes_dir, const char *stem, uint32_t flags) {
	try {
		return scene_file_load(scenes_dir, stem, flags, "rtmi_scene_file_load_ex");
	} catch (const std::exception &e) { // no exception crosses the C boundary
		fail(std::string("rtmi_scene_file_load_ex: ") + e.what());
		return nullptr;
	}
}

rtmi_scene_file *rtmi_scene_file_load(const char *scenes_dir, const char *stem) {
	try {
		return scene_file_load(scenes_dir, stem, 0u, "rtmi_scene_file_load");
	} catch (const std::exception &e) {
		fail(std::string("rtmi_scene_file_load: ") + e.what());
		return nullptr;
	}
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
