// Binned-SAH BVH2 builder (host, multi-threaded).  See bvh_build.hpp.
#include "bvh_build.hpp"
#include <string>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <future>
#include <limits>

namespace rtmi {
namespace {

struct TriBox { float lo[3], hi[3], c[3]; };

inline float round_down(double x) {
	float f = (float)x;
	if ((double)f > x) f = std::nextafterf(f, -std::numeric_limits<float>::infinity());
	return f;
}
inline float round_up(double x) {
	float f = (float)x;
	if ((double)f < x) f = std::nextafterf(f, std::numeric_limits<float>::infinity());
	return f;
}

struct Box {
	float lo[3], hi[3];
	void reset() {
		for (int a = 0; a < 3; ++a) { lo[a] = std::numeric_limits<float>::max(); hi[a] = -std::numeric_limits<float>::max(); }
	}
	void grow(const float *l, const float *h) {
		for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], l[a]); hi[a] = std::max(hi[a], h[a]); }
	}
	void grow(const Box &b) { grow(b.lo, b.hi); }
	float half_area() const {
		float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
		return dx * dy + dy * dz + dz * dx;
	}
};

#ifndef RTMI_NBINS
#define RTMI_NBINS 16
#endif
constexpr int NBINS = RTMI_NBINS;

struct Builder {
	const std::vector<TriBox> &tb;
	std::vector<BuildNode> &nodes;
	std::vector<uint32_t> &order;
	std::atomic<uint32_t> next_node{0};
	std::atomic<uint32_t> depth{0}, leaves{0}, max_leaf_seen{0};
	std::atomic<int> spare_threads;
	uint32_t max_leaf, max_depth;

	Builder(const std::vector<TriBox> &t, std::vector<BuildNode> &n, std::vector<uint32_t> &o, uint32_t ml, uint32_t md,
	        int threads)
	    : tb(t), nodes(n), order(o), spare_threads(threads - 1), max_leaf(ml), max_depth(md) {}

	static void amax(std::atomic<uint32_t> &a, uint32_t v) {
		uint32_t cur = a.load();
		while (cur < v && !a.compare_exchange_weak(cur, v)) {}
	}

	// smallest d with max_leaf·2^d >= count
	uint32_t balanced_depth(uint32_t count) const {
		uint32_t d = 0;
		uint64_t cap = max_leaf;
		while (cap < count) { cap <<= 1; ++d; }
		return d;
	}

	void build(uint32_t id, uint32_t first, uint32_t count, uint32_t d) {
		BuildNode &nd = nodes[id];
		Box b, cb;
		b.reset();
		cb.reset();
		for (uint32_t i = first; i < first + count; ++i) {
			const TriBox &t = tb[order[i]];
			b.grow(t.lo, t.hi);
			cb.grow(t.c, t.c);
		}
		std::memcpy(nd.lo, b.lo, sizeof b.lo);
		std::memcpy(nd.hi, b.hi, sizeof b.hi);
		nd.first = first;
		nd.count = count;
		if (count <= max_leaf) {
			nd.left = nd.right = -1;
			amax(depth, d);
			amax(max_leaf_seen, count);
			leaves.fetch_add(1);
			return;
		}
		uint32_t mid = first;
		// Wall peeling.  Triangles that lie IN one of the node's six boundary planes (a room's floor, a table top, the ground
		// under a scene) are split off as a flat child when they are a sizeable part of the node.  The area heuristic below
		// never does that — it prices a half-room (pieces of five walls in one hollow box) below "one wall | the rest" —
		// yet rays that cross a room's interior then meet hollow box after hollow box; with the walls peeled the top of the
		// hierarchy is six flat boxes a crossing ray misses at once (form factors: DESIGN.md §4, the pair-level beam test).
		if (count >= 64) {
			uint32_t in_plane[6] = {0, 0, 0, 0, 0, 0};
			for (uint32_t i = first; i < first + count; ++i) {
				const TriBox &t = tb[order[i]];
				for (int a = 0; a < 3; ++a) {
					if (t.lo[a] != t.hi[a]) continue;
					if (t.lo[a] == b.lo[a]) ++in_plane[2 * a];
					else if (t.lo[a] == b.hi[a]) ++in_plane[2 * a + 1];
				}
			}
			int pick = -1;
			for (int k = 0; k < 6; ++k)
				if (in_plane[k] < count && in_plane[k] * 8u >= count && (pick < 0 || in_plane[k] > in_plane[pick])) pick = k;
			if (pick >= 0) {
				const int a = pick >> 1;
				const float plane = (pick & 1) ? b.hi[a] : b.lo[a];
				auto it = std::partition(order.begin() + first, order.begin() + first + count,
				                         [&](uint32_t id2) { return tb[id2].lo[a] == plane && tb[id2].hi[a] == plane; });
				mid = (uint32_t)(it - order.begin());
			}
		}
		if (mid == first) {
			int best_axis = -1, best_split = -1;
			float best_cost = std::numeric_limits<float>::max();
			for (int a = 0; a < 3; ++a) {
				float ext = cb.hi[a] - cb.lo[a];
				if (!(ext > 0.f)) continue;
				float scale = NBINS / ext;
				Box bb[NBINS];
				uint32_t cnt[NBINS] = {0};
				for (auto &x : bb) x.reset();
				for (uint32_t i = first; i < first + count; ++i) {
					const TriBox &t = tb[order[i]];
					int k = std::min(NBINS - 1, (int)((t.c[a] - cb.lo[a]) * scale));
					cnt[k]++;
					bb[k].grow(t.lo, t.hi);
				}
				float ra[NBINS];
				uint32_t rc[NBINS];
				Box acc;
				acc.reset();
				uint32_t c = 0;
				for (int k = NBINS - 1; k > 0; --k) {
					if (cnt[k]) acc.grow(bb[k]);
					c += cnt[k];
					ra[k] = c ? acc.half_area() : 0.f;
					rc[k] = c;
				}
				acc.reset();
				c = 0;
				for (int k = 0; k < NBINS - 1; ++k) {
					if (cnt[k]) acc.grow(bb[k]);
					c += cnt[k];
					if (c == 0 || rc[k + 1] == 0) continue;
					float cost = acc.half_area() * (float)c + ra[k + 1] * (float)rc[k + 1];
					if (cost < best_cost) { best_cost = cost; best_axis = a; best_split = k; }
				}
			}
			if (best_axis >= 0) {
				float ext = cb.hi[best_axis] - cb.lo[best_axis];
				float scale = NBINS / ext, base = cb.lo[best_axis];
				int a = best_axis, s = best_split;
				auto it = std::partition(order.begin() + first, order.begin() + first + count, [&](uint32_t id2) {
					return std::min(NBINS - 1, (int)((tb[id2].c[a] - base) * scale)) <= s;
				});
				mid = (uint32_t)(it - order.begin());
			}
		}
		// Depth guarantee: a child with c triangles at depth d+1 must still fit a balanced subtree
		// (balanced_depth(c) + d + 1 <= max_depth); the object median always does.
		if (mid != first && mid != first + count &&
		    balanced_depth(std::max(mid - first, first + count - mid)) + d + 1 > max_depth)
			mid = first;
		if (mid == first || mid == first + count) { // no SAH split, or it would bust the depth budget: object median
			int a = 0;
			float e = -1.f;
			for (int k = 0; k < 3; ++k)
				if (cb.hi[k] - cb.lo[k] > e) { e = cb.hi[k] - cb.lo[k]; a = k; }
			mid = first + count / 2;
			std::nth_element(order.begin() + first, order.begin() + mid, order.begin() + first + count,
			                 [&](uint32_t x, uint32_t y) { return tb[x].c[a] < tb[y].c[a]; });
		}
		uint32_t l = next_node.fetch_add(2), r = l + 1;
		nd.left = (int32_t)l;
		nd.right = (int32_t)r;
		uint32_t lc = mid - first, rcnt = first + count - mid;
		if (count > 32768 && spare_threads.fetch_sub(1) > 0) {
			auto fut = std::async(std::launch::async, [&, l, first, lc, d] { build(l, first, lc, d + 1); });
			build(r, mid, rcnt, d + 1);
			fut.get();
			spare_threads.fetch_add(1);
		} else {
			if (count > 32768) spare_threads.fetch_add(1); // undo the failed reservation
			build(l, first, lc, d + 1);
			build(r, mid, rcnt, d + 1);
		}
	}
};

} // namespace

Bvh2 build_bvh2(const double *xyz, uint64_t n, uint32_t max_leaf, uint32_t max_depth, unsigned n_threads) {
	Bvh2 out;
	out.max_leaf = 0;
	if (n == 0) return out;
	std::vector<TriBox> tb(n);
	for (uint64_t i = 0; i < n; ++i) {
		const double *p = xyz + 9 * i;
		for (int a = 0; a < 3; ++a) {
			double lo = std::min(p[a], std::min(p[3 + a], p[6 + a]));
			double hi = std::max(p[a], std::max(p[3 + a], p[6 + a]));
			tb[i].lo[a] = round_down(lo);
			tb[i].hi[a] = round_up(hi);
			tb[i].c[a] = 0.5f * (tb[i].lo[a] + tb[i].hi[a]);
		}
	}
	out.order.resize(n);
	for (uint64_t i = 0; i < n; ++i) out.order[i] = (uint32_t)i;
	out.nodes.resize(2 * n); // a binary tree with >= 1 triangle per leaf has < 2n nodes
	Builder b(tb, out.nodes, out.order, std::max(1u, max_leaf), max_depth, (int)std::max(1u, n_threads));
	b.max_depth = std::max(b.max_depth, b.balanced_depth((uint32_t)n));
	b.next_node = 1;
	b.build(0, 0, (uint32_t)n, 0);
	out.nodes.resize(b.next_node.load());
	out.depth = b.depth.load();
	out.n_leaves = b.leaves.load();
	out.max_leaf = b.max_leaf_seen.load();
	std::memcpy(out.lo, out.nodes[0].lo, sizeof out.lo);
	std::memcpy(out.hi, out.nodes[0].hi, sizeof out.hi);
	return out;
}

int build_wide(const Bvh2 &bvh, Wide4 &out, std::string &err, unsigned n_threads) {
	// A node adopts grandchildren, always opening the inner child with the largest surface area, until it has four
	// children or only leaves.  Inner children first (consecutive node slots), then the leaves (consecutive triangle
	// positions); nodes in depth-first order.
	out.nodes.clear();
	out.build_of.clear();
	out.leaf_pos.assign(bvh.nodes.size(), 0u);
	out.cover.assign(bvh.nodes.size(), 0xffffffffu);
	out.stack_need = 0;
	if (bvh.nodes.empty()) return 0;
	auto area = [&](int32_t i) {
		const BuildNode &x = bvh.nodes[i];
		const float d0 = x.hi[0] - x.lo[0], d1 = x.hi[1] - x.lo[1], d2 = x.hi[2] - x.lo[2];
		return d0 * d1 + d1 * d2 + d0 * d2;
	};
	uint32_t next_tri = 0;
	if (bvh.nodes[0].left < 0) {
		if (bvh.nodes[0].count != 1) { err = "a leaf with more than one triangle"; return -1; }
		out.leaf_pos[0] = next_tri++;
		return 0;
	}
	// 1. the layout (sequential, cheap): which binary nodes become 4-wide nodes, their children, node slots, triangle positions
	struct Plan { uint32_t build_id; int32_t kid[4]; uint32_t nk, ni, inner_base, tri_base; };
	std::vector<Plan> plan;
	plan.reserve(bvh.nodes.size() / 3 + 1);
	// Order: the root, its inner children and THEIR inner children take the first slots, level by level (out.n_top of
	// them: the kernels may keep those in LDS); below that depth-first, a node's first child laid out right behind it.
	struct Item { uint32_t build_id, out_id, level; };
	std::vector<Item> todo, later;
	size_t head = 0;
	plan.emplace_back();
	todo.push_back({0u, 0u, 0u});
	bool top_phase = true;
	out.n_top = 1;
	while (true) {
		Item it;
		if (top_phase) {
			if (head == todo.size()) { // the two top levels are laid out: the rest goes depth-first
				top_phase = false;
				out.n_top = (uint32_t)plan.size();
				todo.assign(later.rbegin(), later.rend());
				continue;
			}
			it = todo[head++];
		} else {
			if (todo.empty()) break;
			it = todo.back();
			todo.pop_back();
		}
		const BuildNode &x = bvh.nodes[it.build_id];
		Plan p;
		p.build_id = it.build_id;
		p.kid[0] = x.left; p.kid[1] = x.right; p.kid[2] = p.kid[3] = -1;
		int nk = 2;
		while (nk < 4) {
			int best = -1;
			float ba = -1.f;
			for (int k = 0; k < nk; ++k)
				if (bvh.nodes[p.kid[k]].left >= 0 && area(p.kid[k]) > ba) { ba = area(p.kid[k]); best = k; }
			if (best < 0) break;
			const BuildNode &c = bvh.nodes[p.kid[best]];
			p.kid[best] = c.left;
			p.kid[nk++] = c.right;
		}
		std::stable_partition(p.kid, p.kid + nk, [&](int32_t k) { return bvh.nodes[k].left >= 0; });
		int ni = 0;
		while (ni < nk && bvh.nodes[p.kid[ni]].left >= 0) ++ni;
		p.nk = (uint32_t)nk; p.ni = (uint32_t)ni;
		p.inner_base = (uint32_t)plan.size();
		p.tri_base = next_tri;
		for (int k = 0; k < ni; ++k) plan.emplace_back();
		if (top_phase) {
			for (int k = 0; k < ni; ++k) (it.level < 1 ? todo : later).push_back({(uint32_t)p.kid[k], p.inner_base + (uint32_t)k, it.level + 1u});
		} else {
			for (int k = ni - 1; k >= 0; --k) todo.push_back({(uint32_t)p.kid[k], p.inner_base + (uint32_t)k, it.level + 1u}); // child 0 is laid out next
		}
		for (int k = ni; k < nk; ++k) {
			if (bvh.nodes[p.kid[k]].count != 1) { err = "a leaf with more than one triangle"; return -1; }
			out.leaf_pos[p.kid[k]] = next_tri++;
		}
		plan[it.out_id] = p;
	}
	if (next_tri != bvh.order.size()) { err = "internal error: the 4-wide layout lost triangles"; return -1; }
#if RTMI_FF_WIDE_WALK
	// 1b. where each binary node's subtree sits in the wide hierarchy: a wide node's own binary node is the whole of it; a node
	// the collapse opened on the way (between a wide node's binary node and one of its children) is some of its children
	if (plan.size() >= (1u << 27)) { err = "more than 2^27 4-wide nodes"; return -1; }
	{
		std::vector<int32_t> parent(bvh.nodes.size(), -1);
		for (size_t i = 0; i < bvh.nodes.size(); ++i)
			if (bvh.nodes[i].left >= 0) { parent[bvh.nodes[i].left] = (int32_t)i; parent[bvh.nodes[i].right] = (int32_t)i; }
		for (size_t w = 0; w < plan.size(); ++w) {
			const Plan &p = plan[w];
			out.cover[p.build_id] = (uint32_t)w;
			for (uint32_t k = 0; k < p.nk; ++k)
				for (int32_t m = parent[p.kid[k]]; m >= 0 && (uint32_t)m != p.build_id; m = parent[m]) {
					if (out.cover[m] == 0xffffffffu) out.cover[m] = (uint32_t)w;
					out.cover[m] |= 1u << (27 + k);
				}
		}
	}
#endif
	// 2. the node words (parallel): the node's grid and the children's boxes on it
	out.nodes.resize(plan.size());
	out.build_of.resize(plan.size());
	auto encode = [&](size_t lo_i, size_t hi_i) {
		for (size_t i = lo_i; i < hi_i; ++i) {
			const Plan &p = plan[i];
			const BuildNode &x = bvh.nodes[p.build_id];
			// origin = the node's lo corner, step 2^e with 255·2^e >= extent (e as a signed byte)
			uint32_t eb[3];
			double st[3];
			for (int a = 0; a < 3; ++a) {
				const double ext = (double)x.hi[a] - (double)x.lo[a];
				int e = -126;
				if (ext > 0.0) {
					(void)std::frexp(ext / 255.0, &e); // ext/255 = m·2^e, m in [0.5, 1)
					e = std::max(e - 1, -126);
				}
				while (e < 127 && std::ldexp(255.0, e) < ext) ++e;
				eb[a] = (uint32_t)(e & 0xff);
				st[a] = std::ldexp(1.0, e);
			}
			uint32_t w[6] = {0, 0, 0, 0, 0, 0}; // lo.x, hi.x, lo.y, hi.y, lo.z, hi.z: one byte per child
			for (uint32_t k = 0; k < p.nk; ++k) {
				const BuildNode &c = bvh.nodes[p.kid[k]];
				for (int a = 0; a < 3; ++a) {
					// origin + q·2^e is exact in double (q <= 255): the quantised box contains the fp32 box
					double ql = std::floor(((double)c.lo[a] - (double)x.lo[a]) / st[a]);
					double qh = std::ceil(((double)c.hi[a] - (double)x.lo[a]) / st[a]);
					ql = std::min(255.0, std::max(0.0, ql));
					qh = std::min(255.0, std::max(0.0, qh));
					while (ql > 0.0 && (double)x.lo[a] + ql * st[a] > (double)c.lo[a]) ql -= 1.0;
					while (qh < 255.0 && (double)x.lo[a] + qh * st[a] < (double)c.hi[a]) qh += 1.0;
					w[2 * a] |= (uint32_t)ql << (8 * k);
					w[2 * a + 1] |= (uint32_t)qh << (8 * k);
				}
			}
			auto fbits = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; };
			WideNode g;
			g.w[0] = fbits(x.lo[0]); g.w[1] = fbits(x.lo[1]); g.w[2] = fbits(x.lo[2]);
			g.w[3] = eb[0] | (eb[1] << 8) | (eb[2] << 16) | (p.ni << 24) | (p.nk << 28);
			for (int q = 0; q < 6; ++q) g.w[4 + q] = w[q];
			g.w[10] = p.inner_base;
			g.w[11] = ~((p.tri_base - p.ni) << 3);
			out.nodes[i] = g;
			out.build_of[i] = p.build_id;
		}
	};
	const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>(std::max(1u, n_threads), plan.size() / 4096 + 1));
	if (nt == 1) {
		encode(0, plan.size());
	} else {
		std::vector<std::future<void>> jobs;
		const size_t step = (plan.size() + nt - 1) / nt;
		for (unsigned t = 0; t < nt; ++t) {
			const size_t a0 = std::min(plan.size(), (size_t)t * step), a1 = std::min(plan.size(), a0 + step);
			if (a0 < a1) jobs.push_back(std::async(std::launch::async, encode, a0, a1));
		}
		for (auto &j : jobs) j.get();
	}
	// 3. stack bound: on the way down a node leaves at most n_children − 1 entries behind; children have larger indices
	std::vector<uint32_t> need(plan.size(), 0u);
	for (size_t i = plan.size(); i-- > 0;) {
		uint32_t below = 0;
		for (uint32_t k = 0; k < plan[i].ni; ++k) below = std::max(below, need[plan[i].inner_base + k]);
		need[i] = plan[i].nk - 1u + below;
	}
	out.stack_need = need[0];
	return 0;
}

uint64_t check_wide(const Bvh2 &bvh, const Wide4 &wide) {
	uint64_t bad = 0;
	const uint64_t n = bvh.order.size();
	if (n == 0) return wide.nodes.empty() ? 0 : 1;
	// every leaf of the binary hierarchy owns one position, all positions distinct
	std::vector<uint8_t> used(n, 0);
	std::vector<int32_t> leaf_at(n, -1); // position -> build node
	for (size_t i = 0; i < bvh.nodes.size(); ++i)
		if (bvh.nodes[i].left < 0) {
			const uint32_t pos = wide.leaf_pos[i];
			if (pos >= n || used[pos]) { ++bad; continue; }
			used[pos] = 1;
			leaf_at[pos] = (int32_t)i;
		}
	for (uint64_t k = 0; k < n; ++k) bad += used[k] ? 0 : 1;
	if (wide.nodes.empty()) return bad + (bvh.nodes[0].left < 0 ? 0 : 1);
	if (wide.build_of.size() != wide.nodes.size()) return bad + 1;
	// walk the wide hierarchy beside the binary one: a wide node stands for a binary inner node
	struct Item { uint32_t wide_id; int32_t build_id; uint32_t above; };
	std::vector<Item> todo = {{0u, 0, 0u}};
	std::vector<uint8_t> seen_pos(n, 0);
	uint64_t visited = 0;
	while (!todo.empty()) {
		const Item it = todo.back();
		todo.pop_back();
		++visited;
		const WideNode &g = wide.nodes[it.wide_id];
		const BuildNode &x = bvh.nodes[it.build_id];
		float org[3];
		std::memcpy(org, g.w, 12);
		const uint32_t ni = (g.w[3] >> 24) & 15u, nk = g.w[3] >> 28;
		if (nk < 2 || nk > 4 || ni > nk) { ++bad; continue; }
		if (it.above + nk - 1 > wide.stack_need) ++bad;
		double st[3];
		for (int a = 0; a < 3; ++a) {
			if (org[a] != x.lo[a]) ++bad;
			st[a] = std::ldexp(1.0, (int)(int8_t)((g.w[3] >> (8 * a)) & 0xffu));
		}
		// a child stands for a descendant of x at most three levels down (the collapse opens at most two nodes in a row)
		int32_t cand[14];
		int nc = 0;
		cand[nc++] = x.left; cand[nc++] = x.right;
		for (int q = 0; q < nc && nc + 2 <= 14; ++q)
			if (bvh.nodes[cand[q]].left >= 0) { cand[nc++] = bvh.nodes[cand[q]].left; cand[nc++] = bvh.nodes[cand[q]].right; }
		auto is_near_descendant = [&](int32_t b) {
			for (int q = 0; q < nc; ++q)
				if (cand[q] == b) return true;
			return false;
		};
		for (uint32_t k = 0; k < nk; ++k) {
			double lo[3], hi[3];
			for (int a = 0; a < 3; ++a) {
				lo[a] = (double)org[a] + (double)((g.w[4 + 2 * a] >> (8 * k)) & 0xffu) * st[a];
				hi[a] = (double)org[a] + (double)((g.w[5 + 2 * a] >> (8 * k)) & 0xffu) * st[a];
			}
			int32_t child_build = -1;
			if (k < ni) {
				const uint32_t cid = g.w[10] + k;
				if (cid >= wide.nodes.size() || cid <= it.wide_id) { ++bad; continue; }
				const int32_t want = (int32_t)wide.build_of[cid];
				if (is_near_descendant(want) && bvh.nodes[want].left >= 0) child_build = want;
				if (child_build < 0) { ++bad; continue; }
				todo.push_back({cid, child_build, it.above + nk - 1});
			} else {
				const uint32_t ref = g.w[11] - (k << 3); // ~(pos << 3)
				const uint32_t pos = (~ref) >> 3;
				if (((~ref) & 7u) != 0 || pos >= n || leaf_at[pos] < 0 || seen_pos[pos]) { ++bad; continue; }
				seen_pos[pos] = 1;
				child_build = leaf_at[pos];
				if (!is_near_descendant(child_build)) ++bad;
			}
			const BuildNode &c = bvh.nodes[child_build];
			for (int a = 0; a < 3; ++a)
				if (!(lo[a] <= (double)c.lo[a] && hi[a] >= (double)c.hi[a])) ++bad;
		}
	}
	for (uint64_t k = 0; k < n; ++k) bad += seen_pos[k] ? 0 : 1; // every triangle reachable
	if (visited != wide.nodes.size()) ++bad;
#if RTMI_FF_WIDE_WALK
	// cover: the named children of the named wide node are descendants of the binary node, and hold as many triangles as it does
	if (wide.cover.size() != bvh.nodes.size()) return bad + 1;
	std::vector<int32_t> parent(bvh.nodes.size(), -1);
	std::vector<uint32_t> leaves(bvh.nodes.size(), 0u);
	{
		std::vector<int32_t> order; // parents before children
		order.reserve(bvh.nodes.size());
		order.push_back(0);
		for (size_t q = 0; q < order.size(); ++q) {
			const BuildNode &x = bvh.nodes[order[q]];
			if (x.left < 0) continue;
			parent[x.left] = parent[x.right] = order[q];
			order.push_back(x.left);
			order.push_back(x.right);
		}
		for (size_t q = order.size(); q-- > 0;) {
			const BuildNode &x = bvh.nodes[order[q]];
			leaves[order[q]] = x.left < 0 ? 1u : leaves[x.left] + leaves[x.right];
		}
	}
	for (size_t m = 0; m < bvh.nodes.size(); ++m) {
		if (bvh.nodes[m].left < 0) { bad += wide.cover[m] == 0xffffffffu ? 0 : 1; continue; }
		const uint32_t c = wide.cover[m], w = c & 0x7ffffffu, mask = ((c >> 27) & 15u) ? ((c >> 27) & 15u) : 15u;
		if (c == 0xffffffffu || w >= wide.nodes.size()) { ++bad; continue; }
		const WideNode &g = wide.nodes[w];
		const uint32_t ni = (g.w[3] >> 24) & 15u, nk = g.w[3] >> 28;
		uint32_t sum = 0;
		for (uint32_t k = 0; k < nk; ++k) {
			if (!((mask >> k) & 1u)) continue;
			int32_t b = -1;
			if (k < ni) b = g.w[10] + k < wide.build_of.size() ? (int32_t)wide.build_of[g.w[10] + k] : -1;
			else {
				const uint32_t pos = (~(g.w[11] - (k << 3))) >> 3;
				b = pos < n ? leaf_at[pos] : -1;
			}
			if (b < 0) { ++bad; continue; }
			sum += leaves[b];
			int32_t a = b;
			for (int up = 0; up < 4 && a >= 0 && (size_t)a != m; ++up) a = parent[a];
			if (a < 0 || (size_t)a != m) ++bad; // (a child of the whole node's own binary node IS that node's descendant: m == build_of[w] ends the walk too)
		}
		if ((((c >> 27) & 15u) >> nk) != 0u || sum != leaves[m]) ++bad; // (no bit beyond the node's children; 0 = all of them)
	}
#endif
	return bad;
}

} // namespace rtmi
