// Host-only estimate: would spatial splits pay on the C3 soup?  Triangles are PRE-SPLIT (early split clipping:
// a triangle whose box has more surface than a threshold is cut at the middle of the box's longest axis, each part bounded
// by the box of the clipped polygon, recursively) and the library's own binned-SAH builder runs over the parts; a leaf part
// refers to its ORIGINAL triangle, as the kernels' hit ids would.  Counted per C3 primary ray (every 8th pixel): visits of
// binary and 4-wide nodes, exact triangle tests (a triangle met through two of its parts is tested twice: no mailbox on the
// GPU), and the kernel's cost model from round 2's counters (node steps 80 % of the issue slots at 43.4 visits, triangle
// steps 20 % at 3.9 tests: cost = 0.01843·visits + 0.0513·tests, 1.0 = today's kernel).
//   g++ -O2 -std=c++17 -pthread -I raytracer_amd/csrc tools/est_sbvh.cpp raytracer_amd/csrc/bvh_build.cpp -o /tmp/est_sbvh && /tmp/est_sbvh
#include <algorithm>
#include <array>
#include <string>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "bvh_build.hpp"
using namespace rtmi;

struct Ray { double o[3], d[3]; float inv[3]; };
static bool slab(const float *lo, const float *hi, const Ray &r, float tbest, float &tn) {
	float t0 = 0.f, t1 = tbest;
	for (int a = 0; a < 3; ++a) {
		float x = (lo[a] - (float)r.o[a]) * r.inv[a], y = (hi[a] - (float)r.o[a]) * r.inv[a];
		t0 = std::max(t0, std::min(x, y));
		t1 = std::min(t1, std::max(x, y));
	}
	tn = t0;
	return t0 <= t1;
}
static double mt(const double *p, const Ray &r) {
	double e1[3], e2[3], pv[3], tv[3], qv[3];
	for (int a = 0; a < 3; ++a) { e1[a] = p[3 + a] - p[a]; e2[a] = p[6 + a] - p[a]; tv[a] = r.o[a] - p[a]; }
	pv[0] = r.d[1] * e2[2] - r.d[2] * e2[1]; pv[1] = r.d[2] * e2[0] - r.d[0] * e2[2]; pv[2] = r.d[0] * e2[1] - r.d[1] * e2[0];
	double det = e1[0] * pv[0] + e1[1] * pv[1] + e1[2] * pv[2];
	if (std::fabs(det) < 1e-8) return 0;
	double u = (tv[0] * pv[0] + tv[1] * pv[1] + tv[2] * pv[2]) / det;
	if (u < 0 || u > 1) return 0;
	qv[0] = tv[1] * e1[2] - tv[2] * e1[1]; qv[1] = tv[2] * e1[0] - tv[0] * e1[2]; qv[2] = tv[0] * e1[1] - tv[1] * e1[0];
	double v = (r.d[0] * qv[0] + r.d[1] * qv[1] + r.d[2] * qv[2]) / det;
	if (v < 0 || u + v > 1) return 0;
	return (e2[0] * qv[0] + e2[1] * qv[1] + e2[2] * qv[2]) / det;
}

typedef std::vector<std::array<double, 3>> Poly;
static Poly clip(const Poly &in, int axis, double plane, bool keep_below) {
	Poly out;
	for (size_t k = 0; k < in.size(); ++k) {
		const auto &a = in[k], &b = in[(k + 1) % in.size()];
		const bool ia = keep_below ? a[axis] <= plane : a[axis] >= plane, ib = keep_below ? b[axis] <= plane : b[axis] >= plane;
		if (ia) out.push_back(a);
		if (ia != ib) {
			const double t = (plane - a[axis]) / (b[axis] - a[axis]);
			std::array<double, 3> p;
			for (int c = 0; c < 3; ++c) p[c] = a[c] + t * (b[c] - a[c]);
			p[axis] = plane;
			out.push_back(p);
		}
	}
	return out;
}
struct Part { double lo[3], hi[3]; uint32_t orig; };
static void presplit(const Poly &poly, uint32_t orig, double smax, int depth, std::vector<Part> &out) {
	Part p; p.orig = orig;
	for (int a = 0; a < 3; ++a) { p.lo[a] = 1e300; p.hi[a] = -1e300; }
	for (const auto &v : poly) for (int a = 0; a < 3; ++a) { p.lo[a] = std::min(p.lo[a], v[a]); p.hi[a] = std::max(p.hi[a], v[a]); }
	const double d[3] = {p.hi[0] - p.lo[0], p.hi[1] - p.lo[1], p.hi[2] - p.lo[2]};
	const double surf = 2 * (d[0] * d[1] + d[1] * d[2] + d[0] * d[2]);
	if (depth == 0 || surf <= smax || poly.size() < 3) { out.push_back(p); return; }
	const int ax = d[0] >= d[1] && d[0] >= d[2] ? 0 : (d[1] >= d[2] ? 1 : 2);
	const double mid = 0.5 * (p.lo[ax] + p.hi[ax]);
	presplit(clip(poly, ax, mid, true), orig, smax, depth - 1, out);
	presplit(clip(poly, ax, mid, false), orig, smax, depth - 1, out);
}

int main(int argc, char **argv) {
	const uint64_t n = argc > 1 ? strtoull(argv[1], 0, 10) : (1u << 20);
	const double edge = 0.01;
	std::mt19937_64 g(1);
	std::uniform_real_distribution<double> U(-1, 1), E(-edge, edge);
	std::vector<double> xyz(9 * n);
	for (uint64_t i = 0; i < n; ++i) {
		double c[3] = {U(g), U(g), U(g)};
		for (int a = 0; a < 3; ++a) { xyz[9 * i + a] = c[a]; xyz[9 * i + 3 + a] = c[a] + E(g); xyz[9 * i + 6 + a] = c[a] + E(g); }
	}
	// mean surface of a triangle's box, to scale the thresholds
	double mean_surf = 0;
	for (uint64_t i = 0; i < n; ++i) {
		double d[3];
		for (int a = 0; a < 3; ++a) d[a] = std::max({xyz[9 * i + a], xyz[9 * i + 3 + a], xyz[9 * i + 6 + a]}) - std::min({xyz[9 * i + a], xyz[9 * i + 3 + a], xyz[9 * i + 6 + a]});
		mean_surf += 2 * (d[0] * d[1] + d[1] * d[2] + d[0] * d[2]);
	}
	mean_surf /= (double)n;
	printf("%llu triangles, mean box surface %.3g\n", (unsigned long long)n, mean_surf);
	printf("| parts per triangle | threshold (x mean box surface) | binary visits | 4-wide visits | triangle tests | cost model (1.0 = today) |\n|---|---|---|---|---|---|\n");
	for (double f : {1e9, 1.0, 0.5, 0.25, 0.125}) {
		std::vector<Part> parts;
		for (uint64_t i = 0; i < n; ++i) {
			Poly poly(3);
			for (int k = 0; k < 3; ++k) for (int a = 0; a < 3; ++a) poly[k][a] = xyz[9 * i + 3 * k + a];
			presplit(poly, (uint32_t)i, f * mean_surf, 6, parts);
		}
		// the builder takes triangles: hand it, per part, a sliver whose box is the part's box
		std::vector<double> fake(9 * parts.size());
		for (size_t k = 0; k < parts.size(); ++k)
			for (int a = 0; a < 3; ++a) { fake[9 * k + a] = parts[k].lo[a]; fake[9 * k + 3 + a] = parts[k].hi[a]; fake[9 * k + 6 + a] = parts[k].lo[a]; }
		Bvh2 b = build_bvh2(fake.data(), parts.size(), 1, 30, 8);
		// 4-wide collapse as the library does it (largest-surface inner child opened first)
		struct Wide { int n; int child[4]; };
		std::vector<Wide> wide(b.nodes.size());
		auto area = [&](int i) { const BuildNode &x = b.nodes[i]; float d0 = x.hi[0] - x.lo[0], d1 = x.hi[1] - x.lo[1], d2 = x.hi[2] - x.lo[2]; return d0 * d1 + d1 * d2 + d0 * d2; };
		for (size_t i = 0; i < b.nodes.size(); ++i) {
			const BuildNode &x = b.nodes[i];
			if (x.left < 0) continue;
			Wide w; w.n = 2; w.child[0] = x.left; w.child[1] = x.right;
			while (w.n < 4) {
				int best = -1; float ba = -1;
				for (int k = 0; k < w.n; ++k) if (b.nodes[w.child[k]].left >= 0 && area(w.child[k]) > ba) { ba = area(w.child[k]); best = k; }
				if (best < 0) break;
				const BuildNode &c = b.nodes[w.child[best]];
				w.child[best] = c.left; w.child[w.n++] = c.right;
			}
			wide[i] = w;
		}
		const int W = 3840, H = 2160, step = 8;
		unsigned long long v2 = 0, v4 = 0, t2 = 0, t4 = 0, rays = 0;
		for (int py = step / 2; py < H; py += step)
			for (int px = step / 2; px < W; px += step) {
				Ray r;
				r.o[0] = 0; r.o[1] = 0; r.o[2] = -3;
				double x = -1.6 + 3.2 * (px + 0.5) / W, y = 0.9 - 1.8 * (py + 0.5) / H, z = 2.75, l = std::sqrt(x * x + y * y + z * z);
				r.d[0] = x / l; r.d[1] = y / l; r.d[2] = z / l;
				for (int a = 0; a < 3; ++a) r.inv[a] = 1.0f / (float)r.d[a];
				++rays;
				for (int mode = 0; mode < 2; ++mode) {
					double best = 1e300; float tbest = 3e38f;
					struct Ent { int node; float t; };
					Ent stack[256]; int sp = 0; int cur = 0;
					while (true) {
						const BuildNode &nd = b.nodes[cur];
						if (nd.left < 0) {
							for (uint32_t k = 0; k < nd.count; ++k) {
								(mode ? t4 : t2)++;
								double t = mt(&xyz[9 * (uint64_t)parts[b.order[nd.first + k]].orig], r);
								if (t > 0 && t < best) { best = t; tbest = (float)(best * 1.000001); }
							}
						} else {
							(mode ? v4 : v2)++;
							Ent e[4]; int m = 0;
							const int nc = mode ? wide[cur].n : 2;
							for (int k = 0; k < nc; ++k) {
								const int c = mode ? wide[cur].child[k] : (k ? nd.right : nd.left);
								float tn;
								if (slab(b.nodes[c].lo, b.nodes[c].hi, r, tbest, tn)) e[m++] = {c, tn};
							}
							std::sort(e, e + m, [](const Ent &a, const Ent &c) { return a.t > c.t; });
							for (int k = 0; k < m; ++k) stack[sp++] = e[k];
						}
						if (sp == 0) break;
						cur = stack[--sp].node;
					}
				}
			}
		const double R = (double)rays;
		printf("| %.2f | %s | %.1f | %.1f | %.2f (binary %.2f) | %.3f |\n", (double)parts.size() / n, f > 1e8 ? "no splits" : std::to_string(f).c_str(), v2 / R, v4 / R, t4 / R, t2 / R,
		       0.01843 * (v4 / R) + 0.0513 * (t4 / R));
		fflush(stdout);
	}
	return 0;
}
