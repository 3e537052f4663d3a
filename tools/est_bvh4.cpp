// Host-only estimate: node visits and 16-byte requests per C3 primary ray for the binary hierarchy the library builds
// and for its 4-wide collapse, before writing any kernel for the latter.
//   g++ -O2 -std=c++17 -pthread -I raytracer_amd/csrc tools/est_bvh4.cpp raytracer_amd/csrc/bvh_build.cpp -o /tmp/est_bvh4
#include <algorithm>
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

struct Wide { int n; int child[8]; float lo[8][3], hi[8][3]; }; // child: index into bvh.nodes (inner or leaf); boxes as the node stores them
static int g_bits = 0; // 0: the children's fp32 boxes; else planes of g_bits bits relative to the node's box, step a power of two

int main(int argc, char **argv) {
	const uint64_t n = argc > 1 ? strtoull(argv[1], 0, 10) : (1u << 20);
	const int width = argc > 2 ? atoi(argv[2]) : 4;
	const double edge = argc > 3 ? atof(argv[3]) : 0.01;
	g_bits = argc > 4 ? atoi(argv[4]) : 0;
	std::mt19937_64 g(1);
	std::uniform_real_distribution<double> U(-1, 1), E(-edge, edge);
	std::vector<double> xyz(9 * n);
	for (uint64_t i = 0; i < n; ++i) {
		double c[3] = {U(g), U(g), U(g)};
		for (int a = 0; a < 3; ++a) { xyz[9 * i + a] = c[a]; xyz[9 * i + 3 + a] = c[a] + E(g); xyz[9 * i + 6 + a] = c[a] + E(g); }
	}
	Bvh2 b = build_bvh2(xyz.data(), n, 1, 30, 8);
	printf("%llu triangles, %zu build nodes, depth %u\n", (unsigned long long)n, b.nodes.size(), b.depth);
	// collapse: each inner node adopts grandchildren, always opening the inner child with the largest surface area
	std::vector<Wide> wide(b.nodes.size());
	auto area = [&](int i) { const BuildNode &x = b.nodes[i]; float d0 = x.hi[0] - x.lo[0], d1 = x.hi[1] - x.lo[1], d2 = x.hi[2] - x.lo[2]; return d0 * d1 + d1 * d2 + d0 * d2; };
	std::vector<int> todo = {0};
	size_t n_wide = 0, n_children = 0;
	while (!todo.empty()) {
		int i = todo.back(); todo.pop_back();
		const BuildNode &x = b.nodes[i];
		if (x.left < 0) continue;
		Wide w; w.n = 2; w.child[0] = x.left; w.child[1] = x.right;
		while (w.n < width) {
			int best = -1; float ba = -1;
			for (int k = 0; k < w.n; ++k) if (b.nodes[w.child[k]].left >= 0 && area(w.child[k]) > ba) { ba = area(w.child[k]); best = k; }
			if (best < 0) break;
			const BuildNode &c = b.nodes[w.child[best]];
			w.child[best] = c.left; w.child[w.n++] = c.right;
		}
		for (int k = 0; k < w.n; ++k)
			for (int a = 0; a < 3; ++a) {
				const BuildNode &c = b.nodes[w.child[k]];
				if (!g_bits) { w.lo[k][a] = c.lo[a]; w.hi[k][a] = c.hi[a]; continue; }
				const double qmax = (double)((1 << g_bits) - 1), ext = (double)x.hi[a] - x.lo[a];
				int e = -126;
				while (std::ldexp(qmax, e) < ext) ++e;
				const double st = std::ldexp(1.0, e);
				w.lo[k][a] = (float)(x.lo[a] + std::floor(((double)c.lo[a] - x.lo[a]) / st) * st);
				w.hi[k][a] = (float)(x.lo[a] + std::min(qmax, std::ceil(((double)c.hi[a] - x.lo[a]) / st)) * st);
			}
		wide[i] = w; ++n_wide; n_children += w.n;
		for (int k = 0; k < w.n; ++k) todo.push_back(w.child[k]);
	}
	printf("%d-wide: %zu nodes, %.2f children per node\n", width, n_wide, (double)n_children / n_wide);
	// C3 camera, every 8th pixel
	const int W = 3840, H = 2160, step = 8;
	unsigned long long v2 = 0, v4 = 0, t2 = 0, t4 = 0, rays = 0, hits2 = 0, hits4 = 0, boxes4 = 0;
	for (int py = step / 2; py < H; py += step)
		for (int px = step / 2; px < W; px += step) {
			Ray r;
			r.o[0] = 0; r.o[1] = 0; r.o[2] = -3;
			double x = -1.6 + 3.2 * (px + 0.5) / W, y = 0.9 - 1.8 * (py + 0.5) / H, z = 2.75, l = std::sqrt(x * x + y * y + z * z);
			r.d[0] = x / l; r.d[1] = y / l; r.d[2] = z / l;
			for (int a = 0; a < 3; ++a) r.inv[a] = 1.0f / (float)r.d[a];
			++rays;
			for (int mode = 0; mode < 2; ++mode) {
				double best = 1e300; float tbest = 3e38f; bool hit = false;
				struct Ent { int node; float t; };
				Ent stack[256]; int sp = 0; int cur = 0;
				while (true) {
					const BuildNode &x = b.nodes[cur];
					if (x.left < 0) {
						for (uint32_t k = 0; k < x.count; ++k) {
							(mode ? t4 : t2)++;
							double t = mt(&xyz[9 * (uint64_t)b.order[x.first + k]], r);
							if (t > 0 && t < best) { best = t; tbest = (float)(best * 1.000001); hit = true; }
						}
					} else {
						(mode ? v4 : v2)++;
						Ent e[8]; int m = 0;
						Wide w2; w2.n = 2; w2.child[0] = x.left; w2.child[1] = x.right;
						for (int k = 0; k < 2 && !mode; ++k)
							for (int a = 0; a < 3; ++a) { w2.lo[k][a] = b.nodes[w2.child[k]].lo[a]; w2.hi[k][a] = b.nodes[w2.child[k]].hi[a]; }
						const Wide &w = mode ? wide[cur] : w2;
						if (mode) boxes4 += w.n;
						for (int k = 0; k < w.n; ++k) {
							float tn;
							if (slab(w.lo[k], w.hi[k], r, tbest, tn)) e[m++] = {w.child[k], tn};
						}
						std::sort(e, e + m, [](const Ent &a, const Ent &c) { return a.t > c.t; }); // far first
						for (int k = 0; k < m; ++k) stack[sp++] = e[k];
					}
					if (sp == 0) break;
					cur = stack[--sp].node; // entries are not re-tested against tbest on pop, as in the kernel
				}
				(mode ? hits4 : hits2) += hit;
			}
		}
	const double R = (double)rays;
	printf("rays %llu, hits %llu / %llu\n", rays, hits2, hits4);
	printf("binary : %.1f visits, %.2f triangle tests per ray -> %.0f requests (2 per visit, 5 per test), %.0f B of nodes\n", v2 / R, t2 / R,
	       (2.0 * v2 + 5.0 * t2) / R, 32.0 * v2 / R);
	// 16-bit planes on the scene grid: 12 B + 4 B reference per child; b-bit planes relative to the node: 12 B origin,
	// 4 B exponents + leaf mask, 8 B child bases, 6·b/8 B per child
	const int per = g_bits ? (24 + width * 6 * g_bits / 8 + 15) / 16 : (width * 16 + 15) / 16;
	printf("%d-wide : %.1f visits (%.2f boxes each), %.2f triangle tests per ray -> %.0f requests (%d per visit), %.0f B of nodes\n", width, v4 / R,
	       (double)boxes4 / v4, t4 / R, ((double)per * v4 + 5.0 * t4) / R, per, 16.0 * per * v4 / R);
	return 0;
}
