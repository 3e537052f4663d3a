// Binned-SAH BVH2 builder (host, multi-threaded).  See bvh_build.hpp.
#include "bvh_build.hpp"

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

constexpr int NBINS = 16;

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
		{
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

} // namespace rtmi
