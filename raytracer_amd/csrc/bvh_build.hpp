// Host-side BVH construction for librtmi (replaces Scene::BuildBSP, Render2/scene.cpp:463-693).
//
// The reference builds one auto-partitioning BSP per object, splitting triangles into fragments
// with an unseeded RNG.  Any valid bounding-volume hierarchy over the ORIGINAL triangles gives the
// same closest hit, so we build one binned-SAH BVH over all objects' triangles (design nod to the
// reference's dead abvh.cpp: AABB hierarchy + surface-area cost, Render2/abvh.cpp:242-304).
#pragma once
#include <cstdint>
#include <vector>

namespace rtmi {

struct BuildNode {
	float lo[3], hi[3];
	int32_t left = -1, right = -1; // children (index into nodes) or -1 for a leaf
	uint32_t first = 0, count = 0; // leaf: range in `order`
};

struct Bvh2 {
	std::vector<BuildNode> nodes; // nodes[0] = root (if any triangle)
	std::vector<uint32_t> order;  // leaf ranges index this permutation of triangle ids
	uint32_t depth = 0;           // edges on the longest root-to-leaf path
	uint32_t n_leaves = 0;
	uint32_t max_leaf = 0;
	float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
};

// xyz: n × 9 doubles.  Boxes are the triangles' exact bounds rounded OUTWARD to float.
// max_depth bounds the traversal stack: once the remaining depth budget only just fits a balanced
// tree, splits fall back to the object median.
Bvh2 build_bvh2(const double *xyz, uint64_t n, uint32_t max_leaf, uint32_t max_depth, unsigned n_threads);

} // namespace rtmi
